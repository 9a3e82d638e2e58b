// One-off parameter preparation: OIHW -> K-major packed weights, BatchNorm(eval) -> scale/shift.
// (The reference keeps nn.Conv2d / nn.BatchNorm2d parameters as loaded by WeightManager,
//  darknet.py:279-290; these kernels turn them into the layout the conv kernels consume.)
#include "yv3_common.h"

int yv3_pack_weight_planes(const float* w_oihw, void* w_packed, int cout, int cin, int k, int cout_pad, int np, hipStream_t s);

namespace {

template <typename T> __device__ inline T cvt(float v);
template <> __device__ inline float cvt<float>(float v) { return v; }

// out[n][kh][kw][c] = in[n][c][kh][kw]; rows n >= cout are zero-filled.
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ in, T* __restrict__ out,
                                   int cout, int cin, int k, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int kk = k * k;
    const int c = (int)(i % cin);
    long long t = i / cin;
    const int tap = (int)(t % kk);
    const int n = (int)(t / kk);
    float v = 0.f;
    if (n < cout) v = in[((long long)n * cin + c) * kk + tap];
    out[i] = cvt<T>(v);
}

__global__ void fold_bn_kernel(const float* gamma, const float* bias, const float* mean, const float* var,
                               float eps, float* alpha, float* beta, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = gamma[i] / sqrtf(var[i] + eps);
    alpha[i] = a;
    beta[i] = bias[i] - mean[i] * a;
}

}  // namespace

extern "C" int yv3_pack_conv_weight(const float* w_oihw, void* w_packed, int cout, int cin, int k,
                                    int cout_pad, int dtype, void* stream) {
    if (!w_oihw || !w_packed || cout <= 0 || cin <= 0 || cout_pad < cout) return YV3_EINVAL;
    if (k != 1 && k != 3 && k != 4) return YV3_ESHAPE;
    const long long total = (long long)cout_pad * k * k * cin;
    const int blocks = yv3_ceil_div(total, 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == YV3_F32_BF16X3) return yv3_pack_weight_planes(w_oihw, w_packed, cout, cin, k, cout_pad, 3, s);
    if (dtype == YV3_BF16) return yv3_pack_weight_planes(w_oihw, w_packed, cout, cin, k, cout_pad, 1, s);
    if (dtype == YV3_F32_F16X2) return yv3_pack_weight_planes(w_oihw, w_packed, cout, cin, k, cout_pad, 2, s);
    if (dtype == YV3_F32)
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, s, w_oihw, (float*)w_packed, cout, cin, k, total);
    else
        return YV3_EDTYPE;
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_fold_bn(const float* gamma, const float* bias, const float* mean, const float* var,
                           float eps, float* alpha, float* beta, int channels, void* stream) {
    if (!gamma || !bias || !mean || !var || !alpha || !beta || channels <= 0) return YV3_EINVAL;
    hipLaunchKernelGGL(fold_bn_kernel, dim3(yv3_ceil_div(channels, 256)), dim3(256), 0, (hipStream_t)stream,
                       gamma, bias, mean, var, eps, alpha, beta, channels);
    YV3_CHECK_LAUNCH();
    return 0;
}
