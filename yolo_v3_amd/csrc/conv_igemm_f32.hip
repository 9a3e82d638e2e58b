// Fused implicit-GEMM convolution, exact fp32 on the CDNA4 matrix cores.
//
//   y[m][n] = act( sum_k A[m][k] * Wt[n][k] * alpha[n] + beta[n] ) (+ residual[m][n])
//
//   m = (b, ho, wo) output pixel          M = B*Ho*Wo
//   n = output channel                    N = cout
//   k = (kh, kw, cin) K-major             K = k*k*cin
//
// A is never materialised: NHWC activations make every 32-wide K chunk of one output pixel a
// contiguous 128-byte run of the input at pixel (ho*s+kh-pad, wo*s+kw-pad) (or zeros in the
// halo), so a tile row is fetched with eight coalesced 16-byte loads.  Tiles are staged through
// LDS (row pitch 36 floats: conflict-free ds_read_b128 of MFMA fragments) with register prefetch
// of the next K chunk + a double-buffered LDS image, one barrier per chunk.
// Math: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate == an fmaf chain, 157 TFLOP/s peak).
// Epilogue: BN scale/shift (or bias), LeakyReLU(0.1), residual add, straight from the
// accumulators as 128-byte row segments.
//
// Replaces reference darknet.py:43-44 (conv_bn_relu.forward), :52-53 (res_layer.forward),
// :118 (plain head conv) and :161-162 (nearest x2 upsample + cat, folded into the A gather).
#include "yv3_common.h"

namespace {

struct ConvParams {
    const float* x;
    const float* x2;
    const float* w;
    const float* alpha;
    const float* beta;
    const float* res;
    float* y;
    int H, W, Cin, Cup, Cout;
    int stride, act;
    int Ho, Wo, M, K;
    int cchunks;       // Cin / 32
    int nk;            // K / 32
    int ntiles;        // N tiles
};

constexpr int BK = 32;
constexpr int LDS_LD = 36;     // floats per LDS row: 32 + 4 pad -> 144-byte pitch

template <int BM, int BN, int WM, int WN, bool K3, bool DUAL>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvParams p) {
    constexpr int WTM = BM / WM, WTN = BN / WN;      // wave tile
    constexpr int MT = WTM / 32, NT = WTN / 32;      // 32x32 MFMA tiles per wave
    constexpr int AR = BM / 32, BR = BN / 32;        // staging rows per thread
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(MT >= 1 && NT >= 1, "wave tile must hold a 32x32 MFMA tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;       // [2][BN][LDS_LD]

    const int bid = yv3_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % p.ntiles) * BN;
    const int m0 = (bid / p.ntiles) * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int lrow = tid >> 3;      // 0..31
    const int lc4 = tid & 7;        // float4 column within the 32-float chunk

    // ---- per-thread A row descriptors (AR rows, 32 apart)
    long long aoff[AR];             // element offset of (pixel, channel lc4*4) for tap (0,0)
    long long aoff2[DUAL ? AR : 1];
    int ahi[K3 ? AR : 1], awi[K3 ? AR : 1];
    bool aok[AR];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const int m = m0 + lrow + 32 * r;
        aok[r] = m < p.M;
        const int mm = aok[r] ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        if (K3) {
            const int hi0 = ho * p.stride - 1, wi0 = wo * p.stride - 1;
            ahi[r] = hi0; awi[r] = wi0;
            aoff[r] = (((long long)b * p.H + hi0) * p.W + wi0) * p.Cin + lc4 * 4;
        } else if (DUAL) {
            // channels [0,Cup): low-res map [B,H/2,W/2,Cup] at (ho/2, wo/2); rest: x2 [B,H,W,Cin-Cup]
            aoff[r] = (((long long)b * (p.H >> 1) + (ho >> 1)) * (p.W >> 1) + (wo >> 1)) * p.Cup + lc4 * 4;
            aoff2[r] = (((long long)b * p.H + ho) * p.W + wo) * (p.Cin - p.Cup) + lc4 * 4;
        } else {
            aoff[r] = (((long long)b * p.H + ho * p.stride) * p.W + wo * p.stride) * p.Cin + lc4 * 4;
        }
    }
    // ---- per-thread B row offsets
    long long boff[BR];
#pragma unroll
    for (int r = 0; r < BR; ++r) boff[r] = (long long)(n0 + lrow + 32 * r) * p.K + lc4 * 4;

    f32x4 ra[AR], rb[BR];
    int kh = 0, kw = 0, c0 = 0;          // walking (tap, channel) position of the chunk being loaded

    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            bool ok = aok[r];
            const float* src = p.x;
            long long off;
            if (K3) {
                ok = ok && (unsigned)(ahi[r] + kh) < (unsigned)p.H && (unsigned)(awi[r] + kw) < (unsigned)p.W;
                off = aoff[r] + ((long long)kh * p.W + kw) * p.Cin + c0;
            } else if (DUAL) {
                if (c0 < p.Cup) { off = aoff[r] + c0; }
                else { src = p.x2; off = aoff2[r] + (c0 - p.Cup); }
            } else {
                off = aoff[r] + c0;
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(src + off);
            ra[r] = v;
        }
#pragma unroll
        for (int r = 0; r < BR; ++r)
            rb[r] = *reinterpret_cast<const f32x4*>(p.w + boff[r] + (long long)kc * BK);
        // advance the walking position
        c0 += BK;
        if (c0 == p.Cin) { c0 = 0; if (++kw == 3) { kw = 0; ++kh; } }
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * BM * LDS_LD;
        float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int r = 0; r < AR; ++r)
            *reinterpret_cast<f32x4*>(a + (lrow + 32 * r) * LDS_LD + lc4 * 4) = ra[r];
#pragma unroll
        for (int r = 0; r < BR; ++r)
            *reinterpret_cast<f32x4*>(b + (lrow + 32 * r) * LDS_LD + lc4 * 4) = rb[r];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_frag = (wm * WTM + l31) * LDS_LD + lhi * 4;
    const int b_frag = (wn * WTN + l31) * LDS_LD + lhi * 4;

    for (int kc = 0; kc < p.nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < p.nk) load_chunk(kc + 1);
        const float* a = As + buf * BM * LDS_LD + a_frag;
        const float* b = Bs + buf * BN * LDS_LD + b_frag;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < p.nk) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WTN + j * 32 + l31;
        const bool nok = n < p.Cout;
        const float al = (nok && p.alpha) ? p.alpha[n] : 1.f;
        const float be = nok ? p.beta[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (nok && m < p.M) {
                    float v = fmaf(acc[i][j][e], al, be);
                    if (p.act == YV3_ACT_LEAKY) v = v > 0.f ? v : 0.1f * v;
                    const long long o = (long long)m * p.Cout + n;
                    if (p.res) v += p.res[o];
                    p.y[o] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch(const ConvParams& p, bool k3, bool dual, hipStream_t s) {
    const int mtiles = (p.M + BM - 1) / BM;
    const dim3 grid((unsigned)(mtiles * p.ntiles));
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    if (k3)        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), lds, s, p);
    else if (dual) hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, dim3(256), lds, s, p);
    else           hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), lds, s, p);
    YV3_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int yv3_conv2d_f32(const yv3_conv_desc* d, hipStream_t s) {
    ConvParams p;
    p.x = (const float*)d->x; p.x2 = (const float*)d->x2; p.w = (const float*)d->w;
    p.alpha = d->alpha; p.beta = d->beta; p.res = (const float*)d->residual; p.y = (float*)d->y;
    p.H = d->H; p.W = d->W; p.Cin = d->cin; p.Cup = d->cin_up; p.Cout = d->cout;
    p.stride = d->stride; p.act = d->act;
    const int pad = (d->k - 1) / 2;
    p.Ho = (d->H + 2 * pad - d->k) / d->stride + 1;
    p.Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
    const long long M = (long long)d->B * p.Ho * p.Wo;
    if (M > 0x7fffffffLL) return YV3_ESHAPE;
    p.M = (int)M;
    p.K = d->k * d->k * d->cin;
    p.cchunks = d->cin / BK;
    p.nk = p.K / BK;
    const bool k3 = d->k == 3, dual = d->cin_up > 0;

    // Tile selection: widest N tile the layer fills; for launches that would leave most of the
    // 256 CUs idle (small batch at 13x13 / 26x26) fall back to 64x64 tiles for 4x the blocks.
    const int np = d->cout_pad;
    if (np % 128 == 0) {
        const long long blocks128 = ((M + 127) / 128) * (np / 128);
        if (blocks128 >= 384) { p.ntiles = np / 128; return launch<128, 128, 2, 2>(p, k3, dual, s); }
        p.ntiles = np / 64; return launch<64, 64, 2, 2>(p, k3, dual, s);
    }
    if (np % 64 == 0) { p.ntiles = np / 64; return launch<128, 64, 2, 2>(p, k3, dual, s); }
    p.ntiles = np / 32;
    return launch<128, 32, 4, 1>(p, k3, dual, s);
}
