// HBM streaming rate of gfx950 against the bytes a CU keeps in flight: every wave streams 1 KiB pieces (global_load_lds_dwordx4,
// the conv kernels' staging instruction) of a 2 GiB buffer -- far beyond the 256 MB Infinity Cache -- with D pieces outstanding;
// W waves per CU.  Bytes in flight per CU = W x D KiB.  Second table: the same with one 1 KiB store per two loads (the 1x1
// conv layers' read : write mix).  What the short-K 1x1 conv kernels (two 4-wave workgroups per CU, 2-deep ring: ~32 KB of pixel
// rows in flight per CU) can expect from the memory system.
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/hbm_stream tools/probes/hbm_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D, bool WR>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned char* src, unsigned char* dst, long long pieces, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4, g = (long long)blockIdx.x * 4 + wid;
    unsigned char* my = lds + wid * (D * 1024);
    const u32x4 z = {1u, 2u, 3u, 4u};
    long long i = g;
    int slot = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (i < pieces) __builtin_amdgcn_global_load_lds(GPTR(src + i * 1024 + lane * 16), LPTR(my + d * 1024), 16, 0, 0);
        i += nw;
    }
    long long j = g, k = 0;
    for (; i < pieces + nw * D; i += nw) {
        if (WR) {
            // (loads and stores share the vmcnt queue in order: allow the D-1 younger loads + the stores issued since)
            wait_vm<D - 1 + (D - 1) / 2 + 1>();
            if ((k & 1) == 0 && j < pieces) { *reinterpret_cast<u32x4*>(dst + j * 1024 + lane * 16) = z; j += nw; }
            ++k;
        } else wait_vm<D - 1>();
        if (i < pieces) __builtin_amdgcn_global_load_lds(GPTR(src + i * 1024 + lane * 16), LPTR(my + slot * 1024), 16, 0, 0);
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    wait_vm<0>();
    __syncthreads();
    const u32x4 a = *reinterpret_cast<u32x4*>(lds + threadIdx.x * 16);
    if (a[0] == 0x12345678u && a[3] == 7u) out[0] = a[1];
}

template <int D, bool WR>
void run(int wgs, const unsigned char* src, unsigned char* dst, long long pieces, unsigned* dout, int ncu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)4 * D * 1024 > (size_t)(160 * 1024 / wgs) / 1 ? (size_t)4 * D * 1024 : (size_t)4 * D * 1024;
    // occupancy control: request 160 KB / wgs of LDS so that exactly `wgs` workgroups fit a CU
    const size_t req = (size_t)(160 * 1024 / wgs) & ~(size_t)1023;
    const size_t use = req > lds ? req : lds;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<D, WR>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipLaunchKernelGGL((stream_kernel<D, WR>), dim3(ncu * wgs), dim3(256), use, 0, src, dst, pieces / 8, dout);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<D, WR>), dim3(ncu * wgs), dim3(256), use, 0, src, dst, pieces, dout);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double rd = (double)pieces * 1024, wr = WR ? rd / 2 : 0;
    printf("  waves/CU %2d  pieces in flight per wave %2d  = %3d KB per CU : %.3f ms  read %5.2f TB/s%s\n", wgs * 4, D, wgs * 4 * D, best,
           rd / best / 1e9, WR ? "" : "");
    if (WR) printf("      (+ writes: total %5.2f TB/s)\n", (rd + wr) / best / 1e9);
}

int main() {
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const long long bytes = 2LL << 30, pieces = bytes / 1024;
    unsigned char *src, *dst; hipMalloc(&src, bytes); hipMalloc(&dst, bytes / 2 + 65536); hipMemset(src, 1, bytes);
    unsigned* dout; hipMalloc(&dout, 64);
    printf("read only, %d CUs, 2 GiB\n", ncu);
    for (int wgs : {1, 2, 4}) { run<1, false>(wgs, src, dst, pieces, dout, ncu); run<2, false>(wgs, src, dst, pieces, dout, ncu); run<4, false>(wgs, src, dst, pieces, dout, ncu);
                                run<8, false>(wgs, src, dst, pieces, dout, ncu); if (wgs <= 2) run<16, false>(wgs, src, dst, pieces, dout, ncu); }
    printf("two reads : one write\n");
    for (int wgs : {1, 2, 4}) { run<2, true>(wgs, src, dst, pieces, dout, ncu); run<4, true>(wgs, src, dst, pieces, dout, ncu); run<8, true>(wgs, src, dst, pieces, dout, ncu); }
    return 0;
}
