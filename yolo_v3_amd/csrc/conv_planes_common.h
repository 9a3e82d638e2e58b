// Shared pieces of the bf16-plane convolution kernels (conv_planes.hip, conv_planes_k3s1.hip).
#pragma once
#include "yv3_common.h"

struct ConvParamsP {
    const u16* x;
    const u16* x2;
    const u16* w;
    const float* alpha;
    const float* beta;
    const u16* res;
    void* y;
    long long xs, x2s, ys;      // plane strides (elements) of x, x2, y/res
    int H, W, Cin, Cup, Cout;
    int stride, act;
    int Ho, Wo, M, K;
    int nk;                     // K / PBK
    int ntiles;
    int total;                  // workgroup tiles of the launch (m-tiles * ntiles)
    float* dec_out;             // fused YOLO decode of a head conv (yv3_conv_desc.dec_*); NULL: plain logits
    long long dec_bs;
    float dec_stride, dec_an[6]; // anchors already divided by the stride (w0,h0,w1,h1,w2,h2)
    float* ws;                  // stream-K workspace: YV3_SK_MAX_WG accumulator dumps, then YV3_SK_MAX_WG flags
    int* wsflags;
    size_t ws_bytes;
    int tb;                     // rows per packed weight tile
    int* flags;                 // optional: bit 0 <- an fp16-plane output was saturated
    int tune[4];                // tuning experiments (yv3_conv_desc.tune)
    // Winograd launches (conv_planes_kernel<..., WINO>): rows of the GEMM are 2x2 output TILES; position xi's operand
    // matrix starts xi * xi_stride elements into each plane of x; the epilogue maps tile rows back to pixels
    long long xi_stride;
    int wH, wW, wth, wtw;       // output height / width, tile grid
};

// stream-K workspace geometry (yv3_conv_workspace_bytes): one 512-thread workgroup's accumulators per CU + one flag
#define YV3_SK_MAX_WG 512
#define YV3_SK_PART_BYTES (512 * 64 * 4)
// (the Winograd stages' hand-over area -- YV3_WINO_SK_* -- is in yv3_common.h: conv_wino4_f32.hip shares it)

// IO ablations of the epilogue (bit 0 no stores, bit 1 no residual loads, bit 2 no decode arithmetic: results INVALID) exist only in
// measurement builds (-DYV3_MEASURE: `make measure` / tools/build_variant.sh -> libyv3_measure.so / libyv3_<name>.so); the shipped
// libyv3.so ignores yv3_conv_desc.tune[3] -- the branches are compiled out.
#ifdef YV3_MEASURE
#define YV3_IO_ABL(p) ((p).tune[3])
#else
#define YV3_IO_ABL(p) 0
#endif

namespace {

__device__ __attribute__((aligned(64))) u16 g_zero_page[64];     // zero-initialised: source of halo rows


#ifndef YV3_DMA_UNITS
#define YV3_DMA_UNITS 8
#endif
constexpr int PBK = 32;           // K elements per chunk
constexpr int ROWB = PBK * 2;     // bytes per tile row per plane (64: half a cache line)
constexpr int SLOTS = PBK / 8;    // 16-byte slots per row
constexpr int RPG = 64 / SLOTS;   // rows moved by one global_load_lds wave instruction (16)

typedef short bf16x8v __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

// Element type of the planes and the MFMA that consumes them.
//   NP = 1, 3: bf16 (v_cvt_pk_bf16_f32 RN, v_mfma_f32_32x32x16_bf16)
//   NP = 2   : fp16 (v_cvt_f16_f32 RN after saturating to +-65504, v_mfma_f32_32x32x16_f16)
template <int NP> struct PlaneOps {
    static __device__ inline unsigned pack2(float lo, float hi) {
        unsigned r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    }
    static __device__ inline float lo(unsigned q) { return __uint_as_float(q << 16); }
    static __device__ inline float hi(unsigned q) { return __uint_as_float(q & 0xffff0000u); }
    static __device__ inline f32x16 mfma(bf16x8v a, bf16x8v b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __host__ __device__ inline u16 cvt(float v) { return yv3_f2bf(v); }
    static __host__ __device__ inline float back(u16 h) { return yv3_bf2f(h); }
};
template <> struct PlaneOps<2> {
    static __device__ inline _Float16 sat(float v) { return (_Float16)__builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f); }
    static __device__ inline unsigned pack2(float lo, float hi) {       // saturate, then one v_cvt_pk_f16_f32 (RNE)
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 v = {__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f)};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
    }
    static __device__ inline unsigned pack2_nosat(float lo, float hi) {  // for values known to be in range (residues)
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
    }
    static __device__ inline float lo(unsigned q) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(q & 0xffffu)); }
    static __device__ inline float hi(unsigned q) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(q >> 16)); }
    static __device__ inline f32x16 mfma(bf16x8v a, bf16x8v b, f32x16 c) {
#ifdef YV3_EXP_BF16MFMA      // power experiment only (wrong numerics): the same bits through the bf16 multiplier array
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), c, 0, 0, 0);
#endif
    }
    static __device__ inline u16 cvt(float v) { return __builtin_bit_cast(unsigned short, sat(v)); }
    static __device__ inline float back(u16 h) { return (float)__builtin_bit_cast(_Float16, h); }
};

// One "unit" = all partial products of one (weight tile, pixel tile) pair for one 16-deep k-step.
//   NP = 3 (exact bf16 split, 8+8+8 bits): the six products of weight >= 2^-16, smallest first
//   NP = 2 (fp16 split, 11+11(+1) bits)  : w1*x0, w0*x1, w0*x0 (the dropped w1*x1 is <= 2^-22 |w*x|)
//   NP = 1                               : w0*x0
template <int NP>
__device__ inline f32x16 mfma_unit(const bf16x8v* wf, const bf16x8v* xf, f32x16 c) {
    if constexpr (NP == 3) {
        c = PlaneOps<3>::mfma(wf[2], xf[0], c);
        c = PlaneOps<3>::mfma(wf[0], xf[2], c);
        c = PlaneOps<3>::mfma(wf[1], xf[1], c);
        c = PlaneOps<3>::mfma(wf[1], xf[0], c);
        c = PlaneOps<3>::mfma(wf[0], xf[1], c);
    } else if constexpr (NP == 2) {
        c = PlaneOps<2>::mfma(wf[1], xf[0], c);
        c = PlaneOps<2>::mfma(wf[0], xf[1], c);
    }
    return PlaneOps<NP>::mfma(wf[0], xf[0], c);
}

// bank-conflict swizzle for 64-byte rows read with ds_read_b128: four rows share a 256-byte bank row
__device__ __host__ inline int swz(int row) { return (row >> 2) & (SLOTS - 1); }

template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }


// ---- epilogue shared by both kernels: acc[NT][MT] of the wave tile -> BN/activation -> LDS transpose ->
// residual add -> output planes (or fp32).  `lds` must have NW * (MTG*32) * (WTN+4) * 4 bytes available.
// MTG: 32-row accumulator blocks transposed per round (default: the whole wave tile at once).  Wave tiles of 128
// rows (two 4-wave workgroups per CU, 72 KB of LDS each) go through in rounds of MTG = 2 so that the per-wave
// transpose tile fits, and fetch their residual rows round by round (registers).
// WINO: row m of the tile is the 2x2 output tile m of a Winograd launch and `acc` holds its output (wi, wj): the row is
// written to pixel (2 ty + wi, 2 tx + wj) (skipped beyond an odd picture's edge).
template <int NP, int BM, int BN, int WM, int WN, bool OUT_F32, bool SYNC = true, int MTG = BM / WM / 32, bool WINO = false>
__device__ __forceinline__ void epilogue_store(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], const ConvParamsP& p,
                                               unsigned char* lds, int m0, int n0, int wid, int lane, int wi = 0, int wj = 0) {
    static_assert(!(WINO && OUT_F32), "Winograd launches write plane outputs");
    auto rowpix = [&](int m) -> long long {          // pixel index of tile row m, -1: nothing to store
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
    if constexpr (WINO) {                            // the previous output's rows of this wave's LDS tile have been read
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int NG = MT / MTG;                      // rounds
    constexpr int GR = MTG * 32;                      // pixel rows per round
    static_assert(MT % MTG == 0, "rounds must divide the wave tile");
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    (void)NW;
    // ---- epilogue.  D tile of an MFMA: col = lane&31 -> pixel, row = (e&3) + 8*(e>>2) + 4*(lane>>5) -> channel.
    // A lane therefore holds 4 consecutive channels of ONE pixel; storing that directly scatters 8-byte
    // pieces over 32 rows per instruction.  Instead each wave transposes its WTM x WTN tile through LDS
    // (the pipeline stages are free now) so that 8 (or 4) neighbouring lanes cover the contiguous channels
    // of one pixel: residual planes are read and output planes written as full 16-byte-per-lane rows.
    constexpr int EP = WTN + 4;                       // floats per tile row (+4: conflict-free ds_write_b128)
    constexpr int LPR = WTN / 8;                      // lanes per pixel row (8 channels each)
    constexpr int RPP = 64 / LPR;                     // pixel rows per pass
    constexpr int NPASS = GR / RPP;                   // passes per round
    // BN scale / shift of this lane's channels: ALL loads issued back to back, branch-free (clamped indices,
    // out-of-range lanes load valid garbage they never store) -- one L2 round trip instead of one per group.
    const bool has_alpha = p.alpha != nullptr;
    const float* asrc = has_alpha ? p.alpha : p.beta;
    f32x4 alv[NT][4], bev[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * lhi;
            if constexpr (OUT_F32) {          // head conv: cout (255) is not a multiple of 4 -> element-wise
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nn = n + q < p.Cout ? n + q : p.Cout - 1;
                    bev[i][g][q] = p.beta[nn]; alv[i][g][q] = asrc[nn];
                }
            } else {
                const int nn = n < p.Cout ? n : 0;
                bev[i][g] = *reinterpret_cast<const f32x4*>(p.beta + nn);
                alv[i][g] = *reinterpret_cast<const f32x4*>(asrc + nn);
            }
        }
    // Residual rows are fetched next, all passes of a round at once, so that their HBM latency is paid once and overlaps
    // the BN/activation + LDS transpose below (fetching them pass by pass serialises NPASS round trips).
    // NP = 3 would need 96 more registers than the 512-thread kernels have: it keeps the per-pass loads.
    constexpr bool PRE = !OUT_F32 && NP <= 2;
    u32x4 rres[PRE ? NPASS : 1][NP];
    auto fetch_res = [&](int jg) {
        if constexpr (PRE) {
            if (p.res && !(YV3_IO_ABL(p) & 2)) {
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const long long px = rowpix(m0 + wm * WTM + jg * GR + ps * RPP + lane / LPR);
                    const int n = n0 + wn * WTN + (lane % LPR) * 8;
                    const long long o = (px >= 0 && n < p.Cout) ? px * p.Cout + n : 0;      // clamped, unused if out of range
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) rres[ps][pl] = *reinterpret_cast<const u32x4*>(p.res + pl * p.ys + o);
                }
            }
        }
    };
    fetch_res(0);
    // (launch_cfg sizes the dynamic LDS as max(pipeline, NW * GR * EP * 4))
    if (!has_alpha) {                                 // wave-uniform: plain bias convs
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) alv[i][g] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    const float slope = p.act == YV3_ACT_LEAKY ? 0.1f : 1.f;     // LeakyReLU(0.1) == max(t, 0.1 t); linear == max(t, t)
    if constexpr (SYNC) __syncthreads();              // every wave is done with the last stage
    float* tile = reinterpret_cast<float*>(lds) + wid * (GR * EP);
    float amax = 0.f;                                 // running max |v| of what this lane stores (fp16 range check)
    // fused decode state of a head conv (OUT_F32): a lane owns FOUR consecutive channels (fixed over the rows) of one row per pass
    constexpr int LPRF = OUT_F32 ? WTN / 4 : 1;       // lanes per row
    constexpr int RPIF = 64 / LPRF;                   // rows per pass (4 for 64-channel wave tiles)
    const int nd = n0 + wn * WTN + (lane % LPRF) * 4;
    const int attrib = p.Cout / 3;
    int dattr[4];
    float dan[4];
    int db = 0, dgy = 0, dgx = 0;
    if constexpr (OUT_F32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int anc = (nd + q >= attrib) + (nd + q >= 2 * attrib);
            dattr[q] = nd + q - anc * attrib;
            dan[q] = p.dec_an[(anc < 3 ? anc : 2) * 2 + (dattr[q] == 3 ? 1 : 0)];
        }
        // (image, grid y, grid x) of this lane's first row: two divisions once, then carried pass by pass
        if (p.dec_out) {
            const int mf = m0 + wm * WTM + lane / LPRF;
            const int HoWo = p.Ho * p.Wo;
            db = mf / HoWo;
            const int pix = mf - db * HoWo;
            dgy = pix / p.Wo; dgx = pix - dgy * p.Wo;
        }
    }
#pragma unroll
    for (int jg = 0; jg < NG; ++jg) {
        if (jg > 0) {                                 // the previous round's rows have been read by every lane
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            fetch_res(jg);
        }
#pragma unroll
        for (int jj = 0; jj < MTG; ++jj)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int j = jg * MTG + jj;
                    const int nl = i * 32 + 8 * g + 4 * lhi;                 // channel inside the wave tile
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = fmaf(acc[i][j][4 * g + q], alv[i][g][q], bev[i][g][q]);
                        v[q] = __builtin_fmaxf(t, slope * t);
                    }
                    *reinterpret_cast<f32x4*>(tile + (jj * 32 + l31) * EP + nl) = v;
                }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (OUT_F32) {
            // fp32 rows of a head conv (cout = 255: rows are 1020 bytes, consecutive pixels are contiguous, nothing is 16-byte
            // aligned): a lane reads four consecutive channels of a row from the LDS tile (ds_read_b128), decodes them and writes
            // them with ONE 4-byte-aligned 16-byte store, so that a store instruction covers four whole 256-byte row segments; the
            // lane at channel 252 writes three elements (element 255 would be the next pixel's first).
            // Measured (tools/head_probe.py, profiles/r04ad_head_probe.txt): one element per lane and row cost 2x the plane-output
            // epilogue before any decode arithmetic or store.
            static_assert(WTN <= 64 && WTN % 4 == 0 && 64 % LPRF == 0, "whole rows per pass");
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
            float* yf = (float*)p.y;
            const int nv = p.Cout - nd;                                  // valid channels of this lane (>= 4: all)
#pragma unroll 4
            for (int r0 = 0; r0 < GR; r0 += RPIF) {
                const int r = r0 + lane / LPRF;
                const int m = m0 + wm * WTM + jg * GR + r;
                if (m < p.M && nv > 0) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(tile + r * EP + (lane % LPRF) * 4);
                    if (yf) {
                        float* o = yf + (long long)m * p.Cout + nd;
                        if (nv >= 4) *reinterpret_cast<f32x4u*>(o) = t;
                        else { o[0] = t[0]; if (nv > 1) o[1] = t[1]; if (nv > 2) o[2] = t[2]; }
                    }
                    if (p.dec_out) {                                     // fused decode (yv3_decode's map, yololayer.py:31-59,97-105)
                        f32x4 dv;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            dv[q] = (YV3_IO_ABL(p) & 4) ? t[q] : yv3_decode_value(t[q], dattr[q], dan[q], (float)dgx, (float)dgy, p.dec_stride);
                        float* o = p.dec_out + (long long)db * p.dec_bs + (long long)(dgy * p.Wo + dgx) * p.Cout + nd;
                        if (YV3_IO_ABL(p) & 1) asm volatile("" :: "v"(dv));   // (IO ablation for measurements: results INVALID)
                        else if (nv >= 4) *reinterpret_cast<f32x4u*>(o) = dv;
                        else if (nv == 3) *reinterpret_cast<f32x3u*>(o) = f32x3u{dv[0], dv[1], dv[2]};
                        else { o[0] = dv[0]; if (nv > 1) o[1] = dv[1]; }
                    }
                }
                dgx += RPIF;
                while (dgx >= p.Wo) { dgx -= p.Wo; if (++dgy == p.Ho) { dgy = 0; ++db; } }
            }
            continue;
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + lane / LPR;
            const int cg = (lane % LPR) * 8;
            const long long px = rowpix(m0 + wm * WTM + jg * GR + r);
            const int n = n0 + wn * WTN + cg;
            if (px < 0 || n >= p.Cout) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(tile + r * EP + cg);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(tile + r * EP + cg + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const long long o = px * p.Cout + n;
            if (p.res) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {       // planes sum back to the exact fp32 value
                    u32x4 q4;
                    if constexpr (PRE) q4 = rres[ps][pl];
                    else q4 = *reinterpret_cast<const u32x4*>(p.res + pl * p.ys + o);
#pragma unroll
                    for (int h = 0; h < 4; ++h) { v[2 * h] += PlaneOps<NP>::lo(q4[h]); v[2 * h + 1] += PlaneOps<NP>::hi(q4[h]); }
                }
            }
            u16* yo = (u16*)p.y + o;
            if constexpr (NP == 2) {
#pragma unroll
                for (int h = 0; h < 4; ++h) amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v[2 * h]), __builtin_fabsf(v[2 * h + 1])));
                u32x4 qh, ql;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    v[2 * h] = __builtin_amdgcn_fmed3f(v[2 * h], -65504.f, 65504.f);           // saturate (and flag, below)
                    v[2 * h + 1] = __builtin_amdgcn_fmed3f(v[2 * h + 1], -65504.f, 65504.f);
                    qh[h] = PlaneOps<2>::pack2_nosat(v[2 * h], v[2 * h + 1]);
                    ql[h] = PlaneOps<2>::pack2_nosat(v[2 * h] - PlaneOps<2>::lo(qh[h]), v[2 * h + 1] - PlaneOps<2>::hi(qh[h]));
                }
                if (!(YV3_IO_ABL(p) & 1)) {                  // (measurement builds only: IO ablation)
                    *reinterpret_cast<u32x4*>(yo) = qh;
                    *reinterpret_cast<u32x4*>(yo + p.ys) = ql;
                } else { asm volatile("" :: "v"(qh), "v"(ql)); }
            } else {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    u32x4 q4;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        q4[h] = PlaneOps<NP>::pack2(v[2 * h], v[2 * h + 1]);
                        v[2 * h] -= PlaneOps<NP>::lo(q4[h]); v[2 * h + 1] -= PlaneOps<NP>::hi(q4[h]);
                    }
                    *reinterpret_cast<u32x4*>(yo + pl * p.ys) = q4;
                }
            }
        }
    }
    if constexpr (NP == 2) {
        if (p.flags && __any(!(amax <= 65504.f)) && lane == 0) atomicOr(p.flags, 1);
    }
}

// ---- epilogue of a Winograd launch: the tile's FOUR outputs Y[wi][wj] (yac[2 wi + wj]) in one pass structure.  Same arithmetic,
// element by element, as four epilogue_store<..., WINO> calls (bit-identical), but: the BN parameters are loaded once, the tile-row
// -> pixel map (two integer divisions per row) is computed once, and the residual rows of output o+1 are requested while output o
// is processed -- four serialised residual round trips per tile were 6-12 % of a Winograd layer's time
// (profiles/r04p_wino_epilogue_io_ablation.log).  fp16 planes, 32-row wave tiles.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epilogue_store_wino4(f32x16 (&yac)[4][BN / WN / 32][BM / WM / 32], const ConvParamsP& p,
                                                     unsigned char* lds, int m0, int n0, int wid, int lane) {
    constexpr int NP = 2;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int NT = WTN / 32;
    static_assert(WTM == 32, "one 32-row accumulator block per wave");
    constexpr int EP = WTN + 4, LPR = WTN / 8, RPP = 64 / LPR, NPASS = 32 / RPP;
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const bool has_alpha = p.alpha != nullptr;
    const float* asrc = has_alpha ? p.alpha : p.beta;
    f32x4 alv[NT][4], bev[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * lhi;
            const int nn = n < p.Cout ? n : 0;
            bev[i][g] = *reinterpret_cast<const f32x4*>(p.beta + nn);
            alv[i][g] = *reinterpret_cast<const f32x4*>(asrc + nn);
        }
    // tile row -> pixel of output (0, 0), once per pass row; bit 0: row stored at all, bit 1: row 2 ty + 1 inside, bit 2: column 2 tx + 1 inside
    const int ncol = n0 + wn * WTN + (lane % LPR) * 8;
    long long pxb[NPASS];
    int okb[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int m = m0 + wm * WTM + ps * RPP + lane / LPR;
        pxb[ps] = 0; okb[ps] = 0;
        if (m < p.M && ncol < p.Cout) {
            const int tt = p.wth * p.wtw;
            const int b = m / tt;
            const int rem = m - b * tt;
            const int ty = rem / p.wtw, tx = rem - ty * p.wtw;
            pxb[ps] = ((long long)b * p.wH + 2 * ty) * p.wW + 2 * tx;
            okb[ps] = 1 | (2 * ty + 1 < p.wH ? 2 : 0) | (2 * tx + 1 < p.wW ? 4 : 0);
        }
    }
    auto opix = [&](int ps, int o) -> long long {
        const int need = 1 | ((o >> 1) ? 2 : 0) | ((o & 1) ? 4 : 0);
        return (okb[ps] & need) == need ? pxb[ps] + (o >> 1) * p.wW + (o & 1) : -1;
    };
    const bool use_res = p.res && !(YV3_IO_ABL(p) & 2);
    u32x4 rres[2][NPASS][NP];
    auto fetch_res = [&](int o, int buf) {
        if (use_res) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const long long px = opix(ps, o);
                const long long off = px >= 0 ? px * p.Cout + ncol : 0;                    // clamped, unused if out of range
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) rres[buf][ps][pl] = *reinterpret_cast<const u32x4*>(p.res + pl * p.ys + off);
            }
        }
    };
    fetch_res(0, 0);
    if (!has_alpha) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) alv[i][g] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    const float slope = p.act == YV3_ACT_LEAKY ? 0.1f : 1.f;
    float* tile = reinterpret_cast<float*>(lds) + wid * (32 * EP);
    float amax = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        // the previous output's rows of this wave's LDS tile have been read
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = i * 32 + 8 * g + 4 * lhi;
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = fmaf(yac[o][i][0][4 * g + q], alv[i][g][q], bev[i][g][q]);
                    v[q] = __builtin_fmaxf(t, slope * t);
                }
                *reinterpret_cast<f32x4*>(tile + l31 * EP + nl) = v;
            }
        if (o < 3) fetch_res(o + 1, (o + 1) & 1);                 // in flight while this output is split and stored
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + lane / LPR;
            const int cg = (lane % LPR) * 8;
            const long long px = opix(ps, o);
            if (px < 0) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(tile + r * EP + cg);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(tile + r * EP + cg + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const long long off = px * p.Cout + ncol;
            if (p.res) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    const u32x4 q4 = rres[o & 1][ps][pl];
#pragma unroll
                    for (int h = 0; h < 4; ++h) { v[2 * h] += PlaneOps<2>::lo(q4[h]); v[2 * h + 1] += PlaneOps<2>::hi(q4[h]); }
                }
            }
            u16* yo = (u16*)p.y + off;
#pragma unroll
            for (int h = 0; h < 4; ++h) amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v[2 * h]), __builtin_fabsf(v[2 * h + 1])));
            u32x4 qh, ql;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                v[2 * h] = __builtin_amdgcn_fmed3f(v[2 * h], -65504.f, 65504.f);
                v[2 * h + 1] = __builtin_amdgcn_fmed3f(v[2 * h + 1], -65504.f, 65504.f);
                qh[h] = PlaneOps<2>::pack2_nosat(v[2 * h], v[2 * h + 1]);
                ql[h] = PlaneOps<2>::pack2_nosat(v[2 * h] - PlaneOps<2>::lo(qh[h]), v[2 * h + 1] - PlaneOps<2>::hi(qh[h]));
            }
            if (!(YV3_IO_ABL(p) & 1)) {
                *reinterpret_cast<u32x4*>(yo) = qh;
                *reinterpret_cast<u32x4*>(yo + p.ys) = ql;
            } else { asm volatile("" :: "v"(qh), "v"(ql)); }
        }
    }
    if (p.flags && __any(!(amax <= 65504.f)) && lane == 0) atomicOr(p.flags, 1);
}

}  // namespace
