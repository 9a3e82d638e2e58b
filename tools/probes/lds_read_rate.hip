// LDS -> VGPR read rate of one CU on gfx950, as the conv kernels use it: N waves of ONE workgroup per CU (160 KB of LDS requested so
// that no second workgroup shares the CU) issue ds_read_b128 / ds_read_b64 back to back from conflict-free addresses.
// Prints bytes per clock per CU for 1 / 2 / 4 / 8 / 16 waves and three address patterns:
//   linear   lane * 16 (+ 1 KB per instruction)                         -- trivially conflict-free
//   frag64   the MFMA fragment read of conv_planes.hip (64-byte rows, slot ^= (row >> 2) & 3)
//   frag128  the same for 128-byte rows (slot ^= (row >> 1) & 7)
//   bcast    every lane reads the same 16 bytes (LDS broadcast)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/lds_read_rate tools/probes/lds_read_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int PAT, int WIDTH>
__global__ __launch_bounds__(1024) void lds_read_kernel(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 40960; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;     // 160 KB
    __syncthreads();
    int base;
    const int l31 = lane & 31, lhi = lane >> 5;
    if (PAT == 0) base = lane * 16;
    else if (PAT == 1) base = l31 * 64 + ((lhi ^ ((l31 >> 2) & 3)) * 16);
    else if (PAT == 2) base = l31 * 128 + ((lhi ^ ((l31 >> 1) & 7)) * 16);
    else base = 0;
    base += wid * 8192;                                   // each wave its own 8 KB window (16 waves: 128 KB)
    u32x4 acc = {0u, 0u, 0u, 0u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
        // eight reads in flight, then one wait: inline asm so that the loop-invariant loads are neither hoisted nor merged
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = base + (u & 3) * 1024;
            if (WIDTH == 16) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a));
            else { u32x2 w; asm volatile("ds_read_b64 %0, %1" : "=v"(w) : "v"(a)); v[u] = u32x4{w[0], w[1], 0u, 0u}; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wid] = t1 - t0;
    if (acc[0] == 0x12345678u && acc[3] == 7u) out[0] = acc[1] + acc[2];     // keep the reads alive
}

template <int PAT, int WIDTH>
void run(const char* name, int waves, unsigned long long* dout, int ncu) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_read_kernel<PAT, WIDTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((lds_read_kernel<PAT, WIDTH>), dim3(ncu), dim3(64 * waves), 163840, 0, dout, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((lds_read_kernel<PAT, WIDTH>), dim3(ncu), dim3(64 * waves), 163840, 0, dout, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(ncu * 16);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0; int n = 0;
    for (int b = 0; b < ncu; ++b) for (int w = 0; w < waves; ++w) { cyc += (double)h[b * 16 + w]; ++n; }
    cyc /= n;                                             // s_memtime ticks (100 MHz constant clock on gfx9: see the wall-clock column)
    const double bytes = (double)iters * 8 * 64 * WIDTH * waves;                 // per CU
    printf("%-8s b%-3d waves %2d : %8.1f ticks  wall %.3f ms -> %7.1f B/ns/CU  (%.1f B/clk/CU at 2.4 GHz; %5.2f ns per wave-instruction per CU)\n",
           name, WIDTH * 8, waves, cyc, ms, bytes / (ms * 1e6), bytes / (ms * 1e6) / 2.4, ms * 1e6 / ((double)iters * 8 * waves));
}

int main() {
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* dout; hipMalloc(&dout, ncu * 16 * 8);
    for (int waves : {1, 2, 4, 8, 16}) {
        run<0, 16>("linear", waves, dout, ncu);
        run<1, 16>("frag64", waves, dout, ncu);
        run<2, 16>("frag128", waves, dout, ncu);
        run<3, 16>("bcast", waves, dout, ncu);
        run<0, 8>("linear", waves, dout, ncu);
    }
    return 0;
}
