// Issue-rate probe: how many v_mfma_f32_32x32x2_f32 per second does one CU / the chip retire when nothing else happens?
// (4 independent accumulators per wave, operands in registers, W waves per SIMD)   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
}
template <int NACC> void run(int waves_per_simd, int iters) {
    float* out; hipMalloc(&out, 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    const int threads = 64 * 4 * waves_per_simd;          // one workgroup per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(ncu), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(ncu), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)ncu * 4 * waves_per_simd * iters * 16 * NACC;
    const double tf = mfmas * 4096 / (ms * 1e-3) / 1e12;
    printf("acc=%d waves/SIMD=%d : %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NACC, waves_per_simd, ms, tf,
           (ms * 1e-3 * 2.4e9) / ((double)waves_per_simd * iters * 16 * NACC));
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; w *= 2) { run<1>(w, 4000); run<2>(w, 2000); run<4>(w, 1000); }
    return 0;
}
