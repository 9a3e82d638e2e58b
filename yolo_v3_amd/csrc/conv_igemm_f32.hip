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
// LDS (row pitch 36 floats: conflict-free ds_read_b128 of MFMA fragments): two register sets prefetch
// the next TWO K chunks (every load unconditional -> exact vmcnt waits), double-buffered LDS image,
// one barrier per chunk; 128x128 tiles run with eight waves (four per SIMD with two workgroups per CU).
// Math: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate == an fmaf chain, 157 TFLOP/s peak; measured on
// this part with operands in registers and nothing else going on: 156 TFLOP/s on zeros, 136-144 on random
// data, tools/probes/mfma_f32_mix.hip).
// Epilogue: BN scale/shift (or bias), LeakyReLU(0.1), residual add: through LDS as 16-byte row segments
// with the residual reads batched; straight from the accumulators for the 255-channel head convs.
//
// Replaces reference darknet.py:43-44 (conv_bn_relu.forward), :52-53 (res_layer.forward),
// :118 (plain head conv) and :161-162 (nearest x2 upsample + cat, folded into the A gather).
#include <type_traits>
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
    int m_base;        // first output pixel of this launch (rows below it were computed by another launch: conv_gemm_f32.hip)
    // Winograd launches (WINO): rows are 2x2 output tiles, position xi's operand matrix starts xi * xi_stride floats into x
    long long xi_stride;
    int wH, wW, wth, wtw;
};

__device__ __attribute__((aligned(16))) float g_zero_f32[4];      // zero-initialised; NOT const: a constant-address-space pointer would turn the loads into flat_load

constexpr int BK = 32;
constexpr int LDS_LD = 36;     // floats per LDS row: 32 + 4 pad -> 144-byte pitch

// Epilogue through LDS (cout % 4 == 0 and the wave's channel range inside cout): BN scale/shift and LeakyReLU on the
// accumulators, the wave's WTM x WTN tile into its own LDS region, then 16-byte row segments with the residual added --
// residual reads are issued as one batch (the straight-from-accumulator epilogue below issues them one dependent dword at a
// time: ~30 us per 256 x 128 tile).  Same operation order: same bits.
// C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
// WINO: row m = 2x2 output tile m, `acc` = its output (wi, wj) -> pixel (2 ty + wi, 2 tx + wj) (skipped beyond an odd picture's edge)
template <int MT, int NT, bool WINO = false>
__device__ inline void epilogue_lds(const f32x16 (&acc)[MT][NT], const ConvParams& p, float* tile, int mw, int nw, int lane, int wi = 0, int wj = 0) {
    constexpr int WTM = MT * 32, WTN = NT * 32, EP = WTN + 4;
    auto rowpix = [&](int m) -> long long {
        if constexpr (!WINO) return m < p.M ? (long long)m : -1;
        else {
            if (m >= p.M) return -1;
            const int tt = p.wth * p.wtw;
            const int b = m / tt;
            const int rem = m - b * tt;
            const int ty = rem / p.wtw, tx = rem - ty * p.wtw;
            const int oy = 2 * ty + wi, ox = 2 * tx + wj;
            return (oy < p.wH && ox < p.wW) ? ((long long)b * p.wH + oy) * p.wW + ox : -1;
        }
    };
    if constexpr (WINO) {                             // the previous output's rows of this wave's LDS tile have been read
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    constexpr int LPR = WTN / 4, RPP = 64 / LPR, NPASS = WTM / RPP;     // lanes per row, rows per pass, passes
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = nw + j * 32 + l31;
        const float al = p.alpha ? p.alpha[n] : 1.f;
        const float be = p.beta[n];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = fmaf(acc[i][j][e], al, be);
                if (p.act == YV3_ACT_LEAKY) v = v > 0.f ? v : 0.1f * v;
                tile[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi) * EP + j * 32 + l31] = v;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int er = lane / LPR, ec = (lane % LPR) * 4;
    f32x4 rres[NPASS];
    if (p.res) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const long long px = rowpix(mw + ps * RPP + er);
            rres[ps] = *reinterpret_cast<const f32x4*>(p.res + (px >= 0 ? px : 0) * p.Cout + nw + ec);
        }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int r = ps * RPP + er;
        const long long px = rowpix(mw + r);
        f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * EP + ec);
        if (p.res) { v[0] += rres[ps][0]; v[1] += rres[ps][1]; v[2] += rres[ps][2]; v[3] += rres[ps][3]; }
        if (px >= 0) *reinterpret_cast<f32x4*>(p.y + px * p.Cout + nw + ec) = v;
    }
}

// WINO: Winograd F(2x2,3x3) GEMM stage (csrc/winograd.hip): the K loop walks the 16 transform positions (Cin/32 chunks each,
// operand matrix xi * xi_stride into V); at the end of a position the product accumulators are folded into the tile's four
// outputs with the coefficients of A^T x A^T (0 / +-1: exact) and cleared.
// MINW: minimum workgroups per CU the register allocation must allow (the four-wave Winograd tile: 2)
// PIN: the chunk requests stay at the top of the iteration (see there); off under two concurrent lanes
template <int BM, int BN, int WM, int WN, bool K3, bool DUAL, bool WINO = false, int MINW = 1, bool PIN = false>
__global__ __launch_bounds__(64 * WM * WN, MINW) void conv_igemm_f32_kernel(const ConvParams p) {
    static_assert(!WINO || (!K3 && !DUAL), "Winograd stage reads a plain tiles x channels matrix per position");
    constexpr int WTM = BM / WM, WTN = BN / WN;      // wave tile
    constexpr int MT = WTM / 32, NT = WTN / 32;      // 32x32 MFMA tiles per wave
    constexpr int SR = 8 * WM * WN;                  // rows staged per pass: 8 threads (float4 each) per 32-float row
    constexpr int AR = BM / SR, BR = BN / SR;        // staging rows per thread
    static_assert(BM % SR == 0 && BN % SR == 0, "staging passes");
    static_assert(MT >= 1 && NT >= 1, "wave tile must hold a 32x32 MFMA tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;       // [2][BN][LDS_LD]

    const int bid = yv3_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % p.ntiles) * BN;
    const int m0 = (bid / p.ntiles) * BM + p.m_base;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int lrow = tid >> 3;      // 0..SR-1
    const int lc4 = tid & 7;        // float4 column within the 32-float chunk

    // ---- per-thread A row descriptors (AR rows, 32 apart)
    long long aoff[AR];             // element offset of (pixel, channel lc4*4) for tap (0,0)
    long long aoff2[DUAL ? AR : 1];
    int ahi[K3 ? AR : 1], awi[K3 ? AR : 1];
    bool aok[AR];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const int m = m0 + lrow + SR * r;
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
    for (int r = 0; r < BR; ++r) boff[r] = (long long)(n0 + lrow + SR * r) * p.K + lc4 * 4;

    // two register sets: chunk c is requested at the top of iteration c-2 and moved to LDS at the bottom of iteration c-1,
    // so a global load has two compute phases to land
    f32x4 ra[2][AR], rb[2][BR];
    int kh = 0, kw = 0, c0 = 0;          // walking (tap, channel) position of the chunk being loaded

    // Every load is issued unconditionally -- halo / tail rows read a zero page, chunks past the end the zero page and the
    // last weight chunk (never used) -- so each call is exactly AR + BR load instructions and the compiler's s_waitcnt before
    // the LDS write of one staging set is vmcnt(AR + BR), not vmcnt(0): the other, younger set stays in flight.
    auto load_chunk = [&](auto set_, int kc_req) {
        constexpr int S = decltype(set_)::value;
        const bool valid = kc_req < p.nk;
        const int kc = valid ? kc_req : p.nk - 1;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            bool ok = aok[r] && valid;
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
                if (WINO) off += (long long)kw * p.xi_stride;          // (kw counts transform positions)
            }
            ra[S][r] = *reinterpret_cast<const f32x4*>(ok ? src + off : g_zero_f32);
        }
#pragma unroll
        for (int r = 0; r < BR; ++r)
            rb[S][r] = *reinterpret_cast<const f32x4*>(p.w + boff[r] + (long long)kc * BK);
        // advance the walking position
        c0 += BK;
        if (c0 == p.Cin) { c0 = 0; if (WINO) ++kw; else if (++kw == 3) { kw = 0; ++kh; } }
    };
    auto store_chunk = [&](auto set_, int buf) {
        constexpr int S = decltype(set_)::value;
        float* a = As + buf * BM * LDS_LD;
        float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int r = 0; r < AR; ++r)
            *reinterpret_cast<f32x4*>(a + (lrow + SR * r) * LDS_LD + lc4 * 4) = ra[S][r];
#pragma unroll
        for (int r = 0; r < BR; ++r)
            *reinterpret_cast<f32x4*>(b + (lrow + SR * r) * LDS_LD + lc4 * 4) = rb[S][r];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x16 yac[WINO ? 4 : 1][MT][NT];          // WINO: the tile's four outputs Y[wi][wj] (index 2*wi + wj)
    if constexpr (WINO) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) yac[o][i][j][e] = 0.f;
    }
    int wleft = p.cchunks, wxi = 0;

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    load_chunk(S0{}, 0);
    load_chunk(S1{}, 1);
    store_chunk(S0{}, 0);
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_frag = (wm * WTM + l31) * LDS_LD + lhi * 4;
    const int b_frag = (wn * WTN + l31) * LDS_LD + lhi * 4;

    // iteration kc (chunk kc is in LDS buffer kc&1; chunk kc+1 is in flight in register set (kc+1)&1)
    auto iteration = [&](auto par_, int kc) {
        constexpr int P = decltype(par_)::value;                      // kc & 1
        load_chunk(std::integral_constant<int, P>{}, kc + 2);
        // PIN: the requests stay HERE.  Left alone, the scheduler sinks them to the end of the iteration to save registers in the 1x1 and
        // Winograd instantiations, and the next iteration's LDS write then waits (vmcnt 0) for loads issued a few hundred cycles earlier
        // instead of a whole iteration: 416x416 bs=32 +2.4 %, 608x608 bs=16 +2.4 %, bs=1 +5.6 %, bs=64 on one lane +-0; under TWO lanes
        // (the other lane's workgroups already fill such gaps, and the stage then holds 250 instead of 228 registers) -1.0 %: not pinned
        // there (profiles/r05z6_f32_pinned_prefetch_ab.txt).  Scheduling only: same bits.
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        const float* a = As + P * BM * LDS_LD + a_frag;
        const float* b = Bs + P * BN * LDS_LD + b_frag;
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
            // chunk kc+1 (requested 1.5 iterations ago) goes to the other LDS buffer -- free since the last barrier -- in the
            // MIDDLE of this chunk's MFMAs: by the barrier below the writes have long completed (+1 % over writing at the end)
            // (past the end: zeros into a buffer nobody reads)
            if (kk == 1) store_chunk(std::integral_constant<int, 1 - P>{}, 1 - P);
        }
        if constexpr (WINO) {
            if (--wleft == 0) {                                    // end of a transform position: Y += (A^T x A^T)[.][xi] * M
                wleft = p.cchunks;
                const int xr = wxi >> 2, xc = wxi & 3;
                ++wxi;
                const float r0 = xr < 3 ? 1.f : 0.f, r1 = xr == 0 ? 0.f : (xr == 1 ? 1.f : -1.f);
                const float q0 = xc < 3 ? 1.f : 0.f, q1 = xc == 0 ? 0.f : (xc == 1 ? 1.f : -1.f);
                const float sc[4] = {r0 * q0, r0 * q1, r1 * q0, r1 * q1};
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float mv = acc[i][j][e];
#pragma unroll
                            for (int o = 0; o < 4; ++o) yac[o][i][j][e] = fmaf(mv, sc[o], yac[o][i][j][e]);
                            acc[i][j][e] = 0.f;
                        }
            }
        }
        __syncthreads();
    };
    int kc = 0;
    for (; kc + 1 < p.nk; kc += 2) {          // both halves unconditional inside the loop: exact vmcnt bookkeeping
        iteration(S0{}, kc);
        iteration(S1{}, kc + 1);
    }
    if (kc < p.nk) iteration(S0{}, kc);

    // ---- epilogue
    if constexpr (WINO) {                                            // (host: cout % 128 == 0)
#pragma unroll
        for (int o = 0; o < 4; ++o)
            epilogue_lds<MT, NT, true>(yac[o], p, smem + wid * (WTM * (WTN + 4)), m0 + wm * WTM, n0 + wn * WTN, lane, o >> 1, o & 1);
        return;
    }
    if ((p.Cout & 3) == 0 && n0 + wn * WTN + WTN <= p.Cout) {       // wave-uniform; the loop's last __syncthreads freed the LDS
        epilogue_lds<MT, NT>(acc, p, smem + wid * (WTM * (WTN + 4)), m0 + wm * WTM, n0 + wn * WTN, lane);
        return;
    }
    // straight from the accumulators (the 255-channel head convs)
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
int launch(const ConvParams& p, bool k3, bool dual, hipStream_t s, bool pin) {
    const int mtiles = (p.M - p.m_base + BM - 1) / BM;
    const dim3 grid((unsigned)(mtiles * p.ntiles));
    const size_t pipe = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    const size_t epi = (size_t)WM * WN * (BM / WM) * (BN / WN + 4) * sizeof(float);
    const size_t lds = pipe > epi ? pipe : epi;
    const dim3 block(64 * WM * WN);
    if (pin) {
        if (k3)        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, true, false, false, 1, true>), grid, block, lds, s, p);
        else if (dual) hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, true, false, 1, true>), grid, block, lds, s, p);
        else           hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, false, false, 1, true>), grid, block, lds, s, p);
    } else {
        if (k3)        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, block, lds, s, p);
        else if (dual) hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, block, lds, s, p);
        else           hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, block, lds, s, p);
    }
    YV3_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int yv3_wino_input_transform_f32(const float* x, float* v, int B, int H, int W, int C, hipStream_t s);

// Winograd F(2x2,3x3) form of a 3x3 / stride-1 fp32 layer: fp32 MFMA throughout, 2.25x fewer matrix instructions; differs from
// the direct kernel (an fmaf chain in K order) by fp32 round-off of the re-associated sums.
static int launch_wino_f32(const yv3_conv_desc* d, ConvParams p, hipStream_t s) {
    const int th = (d->H + 1) / 2, tw = (d->W + 1) / 2;
    const long long T = (long long)d->B * th * tw;
    if (T > 0x7fffffffLL || d->cout % 128 || d->cout_pad != d->cout || d->cin % 32) return YV3_ESHAPE;
    if (!d->wino_ws || d->wino_ws_bytes < (size_t)16 * T * d->cin * sizeof(float)) return YV3_EWORKSPACE;
    float* v = (float*)d->wino_ws;
    const int rc = yv3_wino_input_transform_f32(p.x, v, d->B, d->H, d->W, d->cin, s);
    if (rc) return rc;
    p.x = v; p.xi_stride = T * d->cin;
    p.w = (const float*)d->w_wino; p.alpha = d->alpha_wino;
    p.wH = d->H; p.wW = d->W; p.wth = th; p.wtw = tw;
    p.H = 1; p.W = (int)T; p.Ho = 1; p.Wo = (int)T; p.M = (int)T; p.stride = 1;
    p.K = 16 * d->cin; p.nk = p.K / BK;
    p.ntiles = d->cout / 128;
    // Round 5: the stage keeps five accumulator sets (216 registers): its 128x128 eight-wave tile runs ONE workgroup per CU, and a launch whose
    // tiles fill e.g. 1.33 rounds of the chip (256->512 @26x26 bs=64: 340 tiles on 256 CUs) spends a whole second tile time on 84 tiles.  The
    // same wave tiles (32x64) as a 64x128 tile on FOUR waves, two workgroups per CU (2 x 55 KB of LDS), halve the scheduling quantum: the last
    // round's half-size tiles spread over more CUs.  Same K order per accumulator: bit-identical.  Chosen when the 128-row tiles leave the last
    // round at most two thirds full beyond the first round, or fill at most half of the chip (then twice as many CUs work).  Same box,
    // alternating (tools/wino_f32_tile_ab.py, profiles/r05t_f32_wino_four_wave_tile_ab.txt): 256->512 @26 bs=64 (340 tiles) 0.615 -> 0.515 ms,
    // 128->256 @52 bs=32 (338) 0.362 -> 0.308, 512->1024 @13 bs=32 (104) 0.549 -> 0.349, @19 bs=16 (104) 0.549 -> 0.349; 200 / 172 / 184 tiles
    // (0.67-0.78 of a round): 2-3 % slower, kept on the eight-wave tile.  (tune[0] == 8 / 9: force the four-wave / the eight-wave tile.)
    const long long t128 = ((T + 127) / 128) * p.ntiles;
    const long long ncu = yv3_num_cu();
    const long long last = t128 % ncu;
    const bool pin = !(d->options & YV3_OPT_TWO_LANES);
    const bool half = d->tune[0] == 8 || (d->tune[0] != 9 && ((t128 > ncu && last > 0 && 3 * last <= 2 * ncu) || 2 * t128 <= ncu));
    if (half) {
        constexpr int BM = 64, BN = 128, WM = 2, WN = 2;
        const dim3 grid((unsigned)(((T + BM - 1) / BM) * p.ntiles));
        const size_t pipe = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
        const size_t epi = (size_t)WM * WN * (BM / WM) * (BN / WN + 4) * sizeof(float);
        if (pin) hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, false, true, 2, true>), grid, dim3(64 * WM * WN), pipe > epi ? pipe : epi, s, p);
        else hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, false, true, 2>), grid, dim3(64 * WM * WN), pipe > epi ? pipe : epi, s, p);
    } else {
        constexpr int BM = 128, BN = 128, WM = 4, WN = 2;
        const dim3 grid((unsigned)(((T + BM - 1) / BM) * p.ntiles));
        const size_t pipe = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
        const size_t epi = (size_t)WM * WN * (BM / WM) * (BN / WN + 4) * sizeof(float);
        if (pin) hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, false, true, 1, true>), grid, dim3(64 * WM * WN), pipe > epi ? pipe : epi, s, p);
        else hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, false, true>), grid, dim3(64 * WM * WN), pipe > epi ? pipe : epi, s, p);
    }
    YV3_CHECK_LAUNCH();
    return 0;
}

int yv3_conv2d_wino4_f32(const yv3_conv_desc* d, hipStream_t s);
int yv3_conv2d_gemm1x1_f32(const yv3_conv_desc* d, hipStream_t s, long long* rows_done);      // csrc/conv_gemm_f32.hip: plain 1x1 layers, persistent DMA-fed GEMM
bool yv3_gemm1x1_f32_takes(const yv3_conv_desc* d);
int yv3_gemm1x1_f32_launches(const yv3_conv_desc* d);
long long yv3_wino4_f32_workgroups(const yv3_conv_desc* d);
bool yv3_wino4_f32_pays(const yv3_conv_desc* d);

// Which form does this fp32 descriptor take: direct (0), Winograd F(2x2,3x3) (1) or F(4x4,3x3) (2)?  (exported through yv3_conv2d_form)
int yv3_conv2d_f32_form(const yv3_conv_desc* d) {
    const bool k3 = d->k == 3, dual = d->cin_up > 0;
    {   // the shape error yv3_conv2d_f32 reports before it launches anything (the form query returns what the launch would)
        const int pad = (d->k - 1) / 2;
        const long long Ho = (d->H + 2 * pad - d->k) / d->stride + 1, Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
        if ((long long)d->B * Ho * Wo > 0x7fffffffLL) return YV3_ESHAPE;
    }
    // F(4x4,3x3) (csrc/conv_wino4_f32.hip): 4x fewer matrix instructions than direct.  Its workgroups are 64 channels x 32 tiles of 4x4 pixels,
    // two per CU: taken by yv3_wino4_f32_pays (tune[0] == 10: never, 11: whenever the filters are there).  Round 6, one item per workgroup, same box,
    // direct / F(2x2) / F(4x4) (profiles/r06o_wino4_forms_by_batch.txt): it is the fastest form of every eligible layer from 8 images of
    // 416x416 up (bs=8: 128->256 @52 0.139 / 0.117 / 0.070 ms, 256->512 @26 0.136 / 0.185 / 0.105; 64 workgroups: 0.187 / 0.337 / 0.182);
    // at 88 workgroups (bs=4 @52) 0.072 / 0.110 / 0.065, at 56 (bs=4 @26) 0.096 / 0.184 / 0.103 -- the crossover
    if (d->w_wino4 && d->wino_ws && k3 && d->stride == 1 && !dual && d->cout % 64 == 0 && (d->cin == 64 || d->cin % 128 == 0) && d->cout_pad == d->cout && d->alpha && d->tune[0] != 10) {
        if ((d->options & YV3_OPT_WINO_ALWAYS) || d->tune[0] == 11 || yv3_wino4_f32_pays(d))
            return d->wino_ws_bytes < yv3_wino4_workspace_bytes(d->B, d->H, d->W, d->cin) ? YV3_EWORKSPACE : YV3_FORM_WINOGRAD4;
    }
    if (!(d->w_wino && d->alpha_wino && k3 && d->stride == 1 && !dual && d->cout % 128 == 0 && d->cout_pad == d->cout)) return 0;
    // fp32 MFMA runs at the vector rate, so this layer is matrix-bound whatever its shape: Winograd whenever the 128x128 tiles
    // (a quarter of the direct kernel's rows) still fill a good part of the chip, or YV3_OPT_WINO_ALWAYS
    const long long T2 = (long long)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2);
    const long long tiles = ((T2 + 127) / 128) * (d->cout / 128);
    if (!((d->options & YV3_OPT_WINO_ALWAYS) || tiles * 100 >= 40 * yv3_num_cu())) return 0;
    return (!d->wino_ws || d->wino_ws_bytes < (size_t)16 * T2 * d->cin * sizeof(float)) ? YV3_EWORKSPACE : YV3_FORM_WINOGRAD;
}

// kernel launches of a descriptor that takes the direct form (exported through yv3_conv2d_launches)
int yv3_conv2d_f32_launches(const yv3_conv_desc* d) { return yv3_gemm1x1_f32_takes(d) ? yv3_gemm1x1_f32_launches(d) : 1; }

int yv3_conv2d_f32(const yv3_conv_desc* d, hipStream_t s) {
    ConvParams p;
    p.x = (const float*)d->x; p.x2 = (const float*)d->x2; p.w = (const float*)d->w;
    p.alpha = d->alpha; p.beta = d->beta; p.res = (const float*)d->residual; p.y = (float*)d->y;
    p.H = d->H; p.W = d->W; p.Cin = d->cin; p.Cup = d->cin_up; p.Cout = d->cout;
    p.stride = d->stride; p.act = d->act; p.m_base = 0;
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
    const bool pin = !(d->options & YV3_OPT_TWO_LANES);           // (see PIN)
    const int form = yv3_conv2d_f32_form(d);
    if (form < 0) return form;
    if (form == YV3_FORM_WINOGRAD4) return yv3_conv2d_wino4_f32(d, s);
    if (form == YV3_FORM_WINOGRAD) return launch_wino_f32(d, p, s);
    if (yv3_gemm1x1_f32_takes(d)) {
        // plain 1x1 / 3x3 layer: whole rounds of the chip on the persistent GEMM, the rest (< half a round of its tiles) on the tiles below
        // (same K order per output element: same bits whoever computes a row)
        long long done = 0;
        const int rc = yv3_conv2d_gemm1x1_f32(d, s, &done);
        if (rc || done >= M) return rc;
        p.m_base = (int)done;
        const int np1 = d->cout_pad;
        if (k3) { p.ntiles = np1 / 128; return launch<128, 128, 4, 2>(p, true, false, s, pin); }
        p.ntiles = np1 / 64;
        return np1 % 128 == 0 ? launch<64, 64, 2, 2>(p, false, false, s, pin) : launch<128, 64, 2, 2>(p, false, false, s, pin);
    }

    // Tile selection: widest N tile the layer fills; for launches that would leave most of the
    // 256 CUs idle (small batch at 13x13 / 26x26) fall back to 64x64 tiles for 4x the blocks.
    const int np = d->cout_pad;
    if (np % 128 == 0) {
        const long long blocks128 = ((M + 127) / 128) * (np / 128);
        // (tune[0]: kernel-selection override for A/B measurements -- 6 four-wave 128x128, 2 64x64 tiles)
        if (blocks128 >= 384 && d->tune[0] == 6) { p.ntiles = np / 128; return launch<128, 128, 2, 2>(p, k3, dual, s, pin); }
        // eight waves (4 x 2 of 32x64) per 128x128 tile, two workgroups per CU: four waves per SIMD hide each other's fragment
        // reads / barriers better than two (13x13 3x3 layer at bs=64: 80 -> 102 TFLOP/s, whole network +5 %)
        // 1x1 layers (K <= 1024: 8-32 chunks per tile) run better on 64x64 tiles, four workgroups per CU: 512->256 @26x26 at bs=64
        // 82 -> 103 TFLOP/s, 256->128 @52x52 95 -> 98 (tune[0] == 7: 128x128 tiles for them too)
        if (blocks128 >= 384 && d->tune[0] != 2 && (k3 || d->tune[0] == 7)) { p.ntiles = np / 128; return launch<128, 128, 4, 2>(p, k3, dual, s, pin); }
        p.ntiles = np / 64; return launch<64, 64, 2, 2>(p, k3, dual, s, pin);
    }
    if (np % 64 == 0) { p.ntiles = np / 64; return launch<128, 64, 2, 2>(p, k3, dual, s, pin); }
    p.ntiles = np / 32;
    return launch<128, 32, 4, 1>(p, k3, dual, s, pin);
}
