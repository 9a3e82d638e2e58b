// First layer (feature.mlist.0, reference darknet.py:76): Conv2d(3->32, 3x3, s1, pad 1, no bias)
// + BatchNorm(eval) + LeakyReLU(0.1).  K = 27 is too shallow for the implicit-GEMM tiles and the
// layer is HBM-bound (AI ~ 12 FLOP/B: 12 B/pixel in, 128 B/pixel out), so it is a direct
// convolution on the vector ALUs:
//   - reads the caller's NCHW fp32 batch directly (lanes = consecutive x: coalesced plane reads,
//     the 3x3 taps overlap in L1), which also folds the NCHW -> NHWC layout change into the layer;
//   - one pixel per lane, 32 accumulators, weights [tap][32] fetched with scalar loads
//     (wave-uniform addresses) and fed to v_fmac as SGPR operands;
//   - the 256x32 output tile is transposed through LDS so every store instruction writes a
//     contiguous run of the NHWC tensor.
#include <stdlib.h>
#include "yv3_common.h"

namespace {

// NP = 0: fp32 NHWC output; NP = 1 / 3: bf16 planes [NP][B,H,W,32] (3 = exact split of the fp32 value);
// NP = 2: fp16 planes (hi + lo)
template <int NP>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                   const float* __restrict__ alpha, const float* __restrict__ beta,
                                                   void* __restrict__ yv, int H, int W, long long plane_stride) {
    __shared__ float tile[256 * 33];
    const int b = blockIdx.y;
    const int HW = H * W;
    const int pix0 = blockIdx.x * 256;
    const int pix = pix0 + threadIdx.x;
    const bool live = pix < HW;
    const int h = live ? pix / W : 0;
    const int w = live ? pix - h * W : 0;

    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = 0.f;

    const float* xb = x + (size_t)b * 3 * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = h + kh - 1;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = w + kw - 1;
                float v = 0.f;
                if (live && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                    v = xb[(size_t)c * HW + (size_t)hi * W + wi];
                const float* wr = wt + ((c * 3 + kh) * 3 + kw) * 32;     // wave-uniform -> s_load
#pragma unroll
                for (int n = 0; n < 32; ++n) acc[n] = fmaf(v, wr[n], acc[n]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 32; ++n) {
        float v = fmaf(acc[n], alpha[n], beta[n]);
        v = v > 0.f ? v : 0.1f * v;
        tile[threadIdx.x * 33 + n] = v;
    }
    __syncthreads();
    const size_t obase = ((size_t)b * HW + pix0) * 32;
    const int nvalid = min(256, HW - pix0) * 32;
    if constexpr (NP == 0) {
        float* y = (float*)yv;
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int e = i * 256 + threadIdx.x;
            if (e < nvalid) y[obase + e] = tile[(e >> 5) * 33 + (e & 31)];
        }
    } else {
        u16* y = (u16*)yv;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int e = 2 * (i * 256 + threadIdx.x);            // two consecutive channels of one pixel
            if (e < nvalid) {
                float v0 = tile[(e >> 5) * 33 + (e & 31)], v1 = tile[(e >> 5) * 33 + (e & 31) + 1];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    u16 h0, h1;
                    if constexpr (NP == 2) {
                        const _Float16 a = (_Float16)v0, b = (_Float16)v1;       // activations of layer 0 are O(1): no saturation needed
                        h0 = __builtin_bit_cast(unsigned short, a); h1 = __builtin_bit_cast(unsigned short, b);
                        v0 -= (float)a; v1 -= (float)b;
                    } else {
                        h0 = yv3_f2bf(v0); h1 = yv3_f2bf(v1);
                        v0 -= yv3_bf2f(h0); v1 -= yv3_bf2f(h1);
                    }
                    *reinterpret_cast<unsigned*>(y + pl * plane_stride + obase + e) = (unsigned)h0 | ((unsigned)h1 << 16);
                }
            }
        }
    }
}

// ---- fp16-plane mode (YV3_F32_F16X2): the layer on the matrix cores.
// K = 27 padded to 32 = two k-steps of v_mfma_f32_32x32x16_f16; operands split hi+lo like every other layer of this
// mode (3 MFMAs per k-step).  A = weights [32 channels x 32 slots], B = pixels [32 slots x 32 pixels], so a lane ends
// up with 16 values of ONE pixel; the channel <-> A-row assignment is permuted so that these are two runs of 8
// consecutive channels (two 16-byte stores per plane, no LDS transpose).  The k-slot <-> tap assignment is chosen so
// that both k-halves of a lane read with the same immediate offsets:
//     k-step 0: lane half h -> channel h, taps j = 0..7 (kh = j/3, kw = j%3)
//     k-step 1: half 0 -> channel 2, taps 0..7;  half 1 -> tap 8 of channels 0,1,2, then 5 zero-weight slots
// A workgroup stages its 10 x 130 x 3 input patch (zero halo) once in LDS, scaled by 16; the weights are scaled by
// 256 (both exact; 2^-12 goes into alpha): the fp16 `lo` parts of inputs >= 2^-6 and weights >= 2^-10 stay normal.
// Values outside the fp16 range after scaling (|x| > 4094, |w| > 255) set bit 0 of *flags.
constexpr int C0_TR = 8, C0_TC = 128;                    // output rows x cols per workgroup
constexpr int C0_PITCH = C0_TC + 4;                      // floats per staged row (130 used)
constexpr int C0_CH = (C0_TR + 2) * C0_PITCH;            // floats per staged channel

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

__device__ inline void split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2v a = {v[2 * q], v[2 * q + 1]};
        const h16x2 h = __builtin_convertvector(a, h16x2);
        const f32x2v r = a - __builtin_convertvector(h, f32x2v);
        const h16x2 l = __builtin_convertvector(r, h16x2);
        hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
    }
}

// BF16_OUT (YV3_BF16 mode): same arithmetic (fp32-class products, fp32 accumulate, as conv0_kernel<1> computes them on the
// vector ALUs), the result rounded to ONE bf16 plane.
template <bool BF16_OUT>
__global__ __launch_bounds__(256) void conv0_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                        const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        u16* __restrict__ y, int H, int W, long long plane_stride,
                                                        int* __restrict__ flags) {
    __shared__ float in[3 * C0_CH];
    const int b = blockIdx.z, r0 = blockIdx.y * C0_TR, c0 = blockIdx.x * C0_TC;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    bool bad = false;

    // ---- stage the input patch: rows r0-1 .. r0+8, cols c0-1 .. c0+128 of the 3 channels, zero outside the image
    const float* xb = x + (size_t)b * 3 * H * W;
    for (int i = tid; i < 3 * (C0_TR + 2) * (C0_TC + 2); i += 256) {
        const int c = i / ((C0_TR + 2) * (C0_TC + 2));
        const int rem = i - c * ((C0_TR + 2) * (C0_TC + 2));
        const int rr = rem / (C0_TC + 2), cc = rem - rr * (C0_TC + 2);
        const int gy = r0 - 1 + rr, gx = c0 - 1 + cc;
        float v = 0.f;
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = xb[((size_t)c * H + gy) * W + gx];
        bad |= !(__builtin_fabsf(v) <= 4094.f);
        in[c * C0_CH + rr * C0_PITCH + cc] = v * 16.f;
    }

    // ---- A operand (weights), once per wave.  Accumulator element e of lane half h is A-row (e&3) + 8*(e>>2) + 4*h;
    // that row carries channel (e&7) + 8*h + 16*(e>>3): the two half-lanes of a pixel then write adjacent 16-byte
    // pieces with each store (channels 0-7 | 8-15, then 16-23 | 24-31)
    const int we = (l31 & 3) + 4 * (l31 >> 3), wh = (l31 >> 2) & 1;
    const int ch = (we & 7) + 8 * wh + 16 * (we >> 3);
    h16x8 whi[2], wlo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = ks == 0 ? lhi * 9 + j : (lhi == 0 ? 18 + j : (j < 3 ? j * 9 + 8 : -1));
            wv[j] = t >= 0 ? wt[t * 32 + ch] * 256.f : 0.f;
            bad |= !(__builtin_fabsf(wv[j]) <= 65504.f);
        }
        split8(wv, whi[ks], wlo[ks]);
    }
    // BN scale/shift of this lane's 16 output channels; 2^-12 undoes the operand scaling
    float al[16], be[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { const int c = (e & 7) + 8 * lhi + 16 * (e >> 3); al[e] = alpha[c] * (1.f / 4096.f); be[e] = beta[c]; }
    // k-step 1 offsets relative to the tap-(0,0) position of channel 0 (see header comment)
    int d1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        d1[j] = lhi == 0 ? 2 * C0_CH + (j / 3) * C0_PITCH + j % 3 : (j % 3) * C0_CH + 2 * C0_PITCH + 2;
    if (flags && __any(bad) && lane == 0) atomicOr(flags, 1);
    __syncthreads();

    // ---- each wave owns a 32-column strip and walks the 8 rows
#pragma unroll 2
    for (int row = 0; row < C0_TR; ++row) {
        const int base = row * C0_PITCH + wid * 32 + l31;
        const float* p0 = in + base + lhi * C0_CH;
        float v0[8], v1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v0[j] = p0[(j / 3) * C0_PITCH + j % 3]; v1[j] = in[base + d1[j]]; }
        h16x8 xhi[2], xlo[2];
        split8(v0, xhi[0], xlo[0]);
        split8(v1, xhi[1], xlo[1]);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[ks], xhi[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[ks], xlo[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[ks], xhi[ks], acc, 0, 0, 0);
        }
        const int gy = r0 + row, gx = c0 + wid * 32 + l31;
        if constexpr (BF16_OUT) {
            if (gy < H && gx < W) {
                u32x4v qb[2];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float t0 = fmaf(acc[2 * q], al[2 * q], be[2 * q]), t1 = fmaf(acc[2 * q + 1], al[2 * q + 1], be[2 * q + 1]);
                    t0 = __builtin_fmaxf(t0, 0.1f * t0); t1 = __builtin_fmaxf(t1, 0.1f * t1);
                    qb[q >> 2][q & 3] = yv3_pack_bf16x2(t0, t1);
                }
                u16* o = y + (((size_t)b * H + gy) * W + gx) * 32 + 8 * lhi;
                *reinterpret_cast<u32x4v*>(o) = qb[0];
                *reinterpret_cast<u32x4v*>(o + 16) = qb[1];
            }
        } else if (gy < H && gx < W) {
            u32x4v qh[2], ql[2];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float t0 = fmaf(acc[2 * q], al[2 * q], be[2 * q]), t1 = fmaf(acc[2 * q + 1], al[2 * q + 1], be[2 * q + 1]);
                t0 = __builtin_fmaxf(t0, 0.1f * t0); t1 = __builtin_fmaxf(t1, 0.1f * t1);          // LeakyReLU(0.1)
                const f32x2v a = {t0, t1};
                const h16x2 h = __builtin_convertvector(a, h16x2);
                const h16x2 l = __builtin_convertvector(a - __builtin_convertvector(h, f32x2v), h16x2);
                qh[q >> 2][q & 3] = __builtin_bit_cast(unsigned, h);
                ql[q >> 2][q & 3] = __builtin_bit_cast(unsigned, l);
            }
            u16* o = y + (((size_t)b * H + gy) * W + gx) * 32 + 8 * lhi;
            *reinterpret_cast<u32x4v*>(o) = qh[0];
            *reinterpret_cast<u32x4v*>(o + 16) = qh[1];
            *reinterpret_cast<u32x4v*>(o + plane_stride) = ql[0];
            *reinterpret_cast<u32x4v*>(o + plane_stride + 16) = ql[1];
        }
    }
}

}  // namespace

// w_tap_major: [c][kh][kw][32] fp32 (27 x 32), i.e. the OIHW weight permuted (1,2,3,0).
// flags (may be NULL): bit 0 is OR-ed in when YV3_F32_F16X2 had to represent a value outside the fp16 range.
extern "C" int yv3_conv0(const float* x_nchw, const float* w_tap_major, const float* alpha, const float* beta,
                         void* y_nhwc, int B, int H, int W, int out_dtype, int* flags, void* stream) {
    if (!x_nchw || !w_tap_major || !alpha || !beta || !y_nhwc || B <= 0 || H <= 0 || W <= 0) return YV3_EINVAL;
    if (out_dtype == YV3_F32_F16X2 || out_dtype == YV3_BF16) {
        const dim3 g((unsigned)yv3_ceil_div(W, C0_TC), (unsigned)yv3_ceil_div(H, C0_TR), (unsigned)B);
        if (out_dtype == YV3_BF16)
            hipLaunchKernelGGL(conv0_mfma_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, x_nchw, w_tap_major, alpha, beta,
                               (u16*)y_nhwc, H, W, (long long)B * H * W * 32, flags);
        else
            hipLaunchKernelGGL(conv0_mfma_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, x_nchw, w_tap_major, alpha, beta,
                               (u16*)y_nhwc, H, W, (long long)B * H * W * 32, flags);
        YV3_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((unsigned)yv3_ceil_div((long long)H * W, 256), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    const long long ps = (long long)B * H * W * 32;
    if (out_dtype == YV3_F32)
        hipLaunchKernelGGL(conv0_kernel<0>, grid, dim3(256), 0, s, x_nchw, w_tap_major, alpha, beta, y_nhwc, H, W, ps);
    else if (out_dtype == YV3_F32_BF16X3)
        hipLaunchKernelGGL(conv0_kernel<3>, grid, dim3(256), 0, s, x_nchw, w_tap_major, alpha, beta, y_nhwc, H, W, ps);
    else
        return YV3_EDTYPE;
    YV3_CHECK_LAUNCH();
    return 0;
}
