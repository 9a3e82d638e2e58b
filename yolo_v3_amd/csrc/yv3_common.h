// Internal helpers shared by the HIP translation units of libyv3.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "yv3.h"

#define YV3_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t _e = hipGetLastError();                   \
        if (_e != hipSuccess) return (int)_e;                \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

static inline int yv3_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Compute units of the CURRENT device, rounded down to a multiple of 8 (equal workgroups per XCD).  Queried per
// call (the runtime serves it from its cached device properties): no process-global state keyed on "whichever
// device launched first".
static inline int yv3_num_cu() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 256;
    return n >= 8 ? n & ~7 : 256;
}

// Hand-over area of the Winograd stages' even schedules (tail of yv3_conv_desc.wino_ws): YV3_WINO_SK_MAX_WG parts, then as many flags.
// A part holds one workgroup's partial outputs: four output accumulator sets of a 512-thread workgroup (F(2x2), fp16 planes: 256 KB) or
// the sixteen outputs of a 256-thread workgroup (F(4x4), exact fp32: 128 KB of it).
#define YV3_WINO_SK_MAX_WG 512
#define YV3_WINO_SK_PART_BYTES (512 * 128 * 4)
static inline size_t yv3_wino_sk_bytes() { return (size_t)YV3_WINO_SK_MAX_WG * (YV3_WINO_SK_PART_BYTES + sizeof(int)) + 256; }

// float -> bf16 bits, round to nearest even (NaN kept quiet)
__host__ __device__ static inline u16 yv3_f2bf(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
// two fp32 -> packed bf16 pair, round to nearest even: the hardware conversion (one instruction instead of ~14; same bits as yv3_f2bf for
// every non-NaN input -- conv_planes_common.h's epilogues use the same instruction)
__device__ static inline unsigned yv3_pack_bf16x2(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__host__ __device__ static inline float yv3_bf2f(u16 h) {
    union { float f; uint32_t u; } v; v.u = ((uint32_t)h) << 16;
    return v.f;
}

// Bijective XCD-aware block remap (MI355X: block b runs on XCD b % 8, each XCD has its own L2).
// Gives every XCD a contiguous range of logical tile ids so neighbouring tiles share an L2.
__device__ static inline int yv3_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// YOLO decode of one logit (reference yololayer.py:31-59,97-105), shared by decode.hip and the fused head-conv
// epilogue so that both produce the same bits.  attr: 0,1 = x,y; 2,3 = w,h; >= 4 = conf / class.
//   xy : (sigmoid(t) + grid) * stride      wh : (exp(t) * (anchor / stride)) * stride      else : sigmoid(t)
// Branch-free (a wave's lanes hold all kinds of attributes: a branch would run both sides) with ONE hardware exponential per
// element: e^u = 2^(u*log2e), the product's rounding error and log2e's tail folded back in as a factor (1 + c*ln2), so that the
// result stays within ~2 ulp of the correctly rounded e^u for every u (v_exp_f32 itself: 1 ulp); sigmoid = v_rcp_f32(1 + e^-t)
// (1 ulp) -- in all <= 4 ulp from the reference's values, inside the 1e-6 relative tolerance the decode tests assert.  The library
// exp + IEEE division this replaces were 30 % of a head conv's time (profiles/r04ad_head_probe.txt).
#ifdef __HIPCC__
__device__ static inline float yv3_exp(float u) {
    // arguments outside [-104, 89] already give 0 / inf in fp32 (e^-104 < the smallest denormal, e^89 > FLT_MAX): clamping there
    // keeps +-inf and |u| > 2.36e38 (whose product with log2e overflows: inf - inf) away from the compensation term, so that
    // exp(+inf) = inf, exp(-inf) = 0, sigmoid(+-inf) = 1 / 0 as the reference's expf / division give.  v_med3_f32
    // returns the other two operands' minimum for a NaN input; the NaN is restored below (exp(NaN) = NaN like the reference).
    const float uc = __builtin_amdgcn_fmed3f(u, -104.0f, 89.0f);
    const float a = uc * 1.44269502e+0f;                                      // log2(e) rounded to float
    const float c = fmaf(uc, 1.92596299e-8f, fmaf(uc, 1.44269502e+0f, -a));   // + its tail, + the product's rounding error
    const float r = __builtin_amdgcn_exp2f(a) * fmaf(c, 0.693147182f, 1.0f);
    return u != u ? u : r;
}
__device__ static inline float yv3_decode_value(float t, int attr, float an, float gx, float gy, float stride) {
#pragma clang fp contract(off)
    const bool wh = attr == 2 || attr == 3;
    const float e = yv3_exp(wh ? t : -t);
    const float sg = __builtin_amdgcn_rcpf(1.f + e);
    const float whv = (e * an) * stride;
    const float xyv = (sg + (attr == 0 ? gx : gy)) * stride;
    return wh ? whv : (attr >= 4 ? sg : xyv);
}
#endif
