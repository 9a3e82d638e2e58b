// Issue-rate probe 2: the compute segment of conv_igemm_f32_pp_kernel in isolation -- 64 v_mfma_f32_32x32x2_f32 over four
// accumulators with 32 + 32 distinct operand registers, 8 waves per CU; optionally the two-group ping-pong barriers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 af[4][2], bf[4][2];
    for (int kk = 0; kk < 4; ++kk) for (int i = 0; i < 2; ++i) {
        af[kk][i] = *reinterpret_cast<const f32x4*>(in + (threadIdx.x * 16 + kk * 2 + i) * 4);
        bf[kk][i] = *reinterpret_cast<const f32x4*>(in + (threadIdx.x * 16 + 8 + kk * 2 + i) * 4);
    }
    const int grp = threadIdx.x >> 8;
    if (MODE == 1 && grp == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i][t], bf[kk][j][t], acc[i][j], 0, 0, 0);
        if (MODE == 1) { __builtin_amdgcn_sched_barrier(0); if (!(grp == 1 && it + 1 == iters)) __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE> void run(int iters, const float* in, float* out) {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(512), 0, 0, out, in, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(512), 0, 0, out, in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)ncu * 8 * iters * 64;
    printf("mode %d (%s): %.3f ms  %.1f TFLOP/s\n", MODE, MODE ? "two groups alternating at s_barrier" : "free-running, 2 waves per SIMD", ms,
           mfmas * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    const int n = 512 * 16 * 4;
    float *in, *out; (void)hipMalloc(&in, n * 4); (void)hipMemset(in, 0, n * 4); (void)hipMalloc(&out, 4);
    printf("operands = 0:\n");
    run<0>(2000, in, out); run<1>(2000, in, out);
    float* h = (float*)malloc(n * 4);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1e-3f; }   // small: no overflow over 1e5 accumulations
    (void)hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    printf("operands = uniform random:\n");
    run<0>(2000, in, out); run<1>(2000, in, out);
    return 0;
}
