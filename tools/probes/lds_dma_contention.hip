// Do L2 -> LDS DMA writes and ds_read_b128 fragment reads compete for the LDS?  One workgroup of 8 waves per CU (160 KB of LDS
// requested): waves 0-3 issue ds_read_b128 batches (8 in flight), waves 4-7 issue global_load_lds_dwordx4 batches (8 in flight) from
// an L2-resident window.  Three runs: reads only, DMA only, both -- per-CU rates and the slowdown of each side when the other runs.
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/lds_dma_contention tools/probes/lds_dma_contention.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mix_kernel(const unsigned char* src, unsigned* out, unsigned long long* ticks, int rd_iters, int dma_iters, int win) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 40960; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    u32x4 acc = {0u, 0u, 0u, 0u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wid < 4) {
        const int l31 = lane & 31, lhi = lane >> 5;
        const int base = 65536 + wid * 8192 + l31 * 64 + ((lhi ^ ((l31 >> 2) & 3)) * 16);      // the conv kernels' fragment pattern
        for (int it = 0; it < rd_iters; ++it) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int a = base + (u & 3) * 2048; asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a)); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    } else {
        const unsigned char* base = src + (size_t)blockIdx.x * win + (wid - 4) * 8192 + (lane >> 2) * 512 + (lane & 3) * 16;   // 16 rows x 64 B
        for (int it = 0; it < dma_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                __builtin_amdgcn_global_load_lds(GPTR(base + (u & 3) * 64), LPTR(lds + (wid - 4) * 8192 + u * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[blockIdx.x * 8 + wid] = t1 - t0;
    if (acc[0] == 0x12345678u && acc[3] == 7u) out[0] = acc[1] + acc[2];
}

static double run(const unsigned char* src, unsigned* dout, unsigned long long* dt, int ncu, int win, int rd, int dma, double* rd_ticks, double* dma_ticks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipLaunchKernelGGL(mix_kernel, dim3(ncu), dim3(512), 163840, 0, src, dout, dt, rd ? 10 : 0, dma ? 10 : 0, win);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mix_kernel, dim3(ncu), dim3(512), 163840, 0, src, dout, dt, rd, dma, win);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long* h = new unsigned long long[ncu * 8];
    hipMemcpy(h, dt, ncu * 8 * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < ncu; ++i) { for (int w = 0; w < 4; ++w) a += h[i * 8 + w]; for (int w = 4; w < 8; ++w) b += h[i * 8 + w]; }
    *rd_ticks = a / (ncu * 4); *dma_ticks = b / (ncu * 4);
    delete[] h;
    return ms;
}

int main() {
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int win = 4 * 8192 + 16384;
    unsigned char* src; hipMalloc(&src, (size_t)ncu * win + 65536); hipMemset(src, 1, (size_t)ncu * win + 65536);
    unsigned* dout; hipMalloc(&dout, 64);
    unsigned long long* dt; hipMalloc(&dt, ncu * 8 * 8);
    const int RD = 4000, DMA = 1000;              // reads: 4 waves x 4000 x 8 KB = 128 MB per CU; DMA: 4 waves x 1000 x 8 KB = 32 MB per CU
    double r, d;
    const double ms_r = run(src, dout, dt, ncu, win, RD, 0, &r, &d);
    printf("reads only : %.3f ms  -> %.1f B/ns/CU read   (s_memtime ticks per read wave %.0f)\n", ms_r, 4.0 * RD * 8192 / (ms_r * 1e6), r);
    const double ms_d = run(src, dout, dt, ncu, win, 0, DMA, &r, &d);
    printf("DMA only   : %.3f ms  -> %.1f B/ns/CU DMA    (ticks per DMA wave %.0f)\n", ms_d, 4.0 * DMA * 8192 / (ms_d * 1e6), d);
    const double ms_b = run(src, dout, dt, ncu, win, RD, DMA, &r, &d);
    const double tick_ns = ms_r * 1e6 / 1.0;      // (ticks are reported raw; wall times give the rates)
    (void)tick_ns;
    printf("both       : %.3f ms  (reads alone %.3f + DMA alone %.3f = %.3f; max %.3f)   ticks: read waves %.0f, DMA waves %.0f\n",
           ms_b, ms_r, ms_d, ms_r + ms_d, ms_r > ms_d ? ms_r : ms_d, r, d);
    printf("=> if the LDS serves both independently, 'both' ~ max; if they share one port, 'both' ~ sum\n");
    return 0;
}
