// L2 -> LDS DMA rate of one CU on gfx950 (global_load_lds_dwordx4, 1 KiB per wave instruction), next to plain
// global_load_dwordx4 into VGPRs: N waves of ONE workgroup per CU (160 KB of LDS requested) stream a buffer that fits L2
// (per CU its own 64 KB window, re-read every iteration) with 8 loads in flight per wave.
//   rows64   lane -> 16 rows x 64 B (the conv kernels' pixel-side pattern: 16 half lines per instruction)
//   rows128  lane ->  8 rows x 128 B (full lines)
//   rows32   lane -> 32 rows x 32 B (round 5: the four-wave kernel's 16-deep chunks)
//   linear   1 KB contiguous (the packed weights' pattern)
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/dma_rate tools/probes/dma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool DMA>
__global__ __launch_bounds__(1024) void dma_kernel(const unsigned char* src, unsigned* out, int iters, int win) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * win + wid * 8192;
    long long off;
    if (PAT == 0) off = (lane >> 2) * 512 + (lane & 3) * 16;            // 16 rows, 512 B apart, 64 B each
    else if (PAT == 1) off = (lane >> 3) * 1024 + (lane & 7) * 16;       // 8 rows, 1 KB apart, 128 B each
    else if (PAT == 3) off = (lane >> 1) * 256 + (lane & 1) * 16;        // 32 rows, 256 B apart, 32 B each (16-deep K chunks: conv_planes_w4.hip)
    else off = lane * 16;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int it = 0; it < iters; ++it) {
        if (DMA) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                __builtin_amdgcn_global_load_lds(GPTR(base + off + (u & 3) * 64), LPTR(lds + wid * 8192 + u * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(base + off + (u & 3) * 64));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    }
    if (DMA) { __syncthreads(); acc = *reinterpret_cast<u32x4*>(lds + tid * 16); }
    if (acc[0] == 0x12345678u && acc[3] == 7u) out[0] = acc[1] + acc[2];
}

template <int PAT, bool DMA>
void run(const char* name, int waves, const unsigned char* src, unsigned* dout, int ncu, int win) {
    const int iters = 1000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<PAT, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((dma_kernel<PAT, DMA>), dim3(ncu), dim3(64 * waves), 163840, 0, src, dout, 20, win);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((dma_kernel<PAT, DMA>), dim3(ncu), dim3(64 * waves), 163840, 0, src, dout, iters, win);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)iters * 8 * 1024 * waves;
    printf("%-8s %-18s waves %2d : wall %.3f ms -> %7.1f B/ns/CU (%.1f B/clk/CU at 2.4 GHz; %6.2f ns per wave-instruction per CU)\n",
           name, DMA ? "global_load_lds x4" : "global_load_dwordx4", waves, ms, bytes / (ms * 1e6), bytes / (ms * 1e6) / 2.4, ms * 1e6 / ((double)iters * 8 * waves));
}

int main() {
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int win = 16 * 8192 + 16384;                   // per CU: 16 waves x 8 KB (+ slack for the row patterns)
    unsigned char* src; hipMalloc(&src, (size_t)ncu * win + 65536); hipMemset(src, 1, (size_t)ncu * win + 65536);
    unsigned* dout; hipMalloc(&dout, 64);
    for (int waves : {1, 2, 4, 8, 16}) {
        run<0, true>("rows64", waves, src, dout, ncu, win);
        run<1, true>("rows128", waves, src, dout, ncu, win);
        run<3, true>("rows32", waves, src, dout, ncu, win);
        run<2, true>("linear", waves, src, dout, ncu, win);
        run<0, false>("rows64", waves, src, dout, ncu, win);
        run<2, false>("linear", waves, src, dout, ncu, win);
    }
    return 0;
}
