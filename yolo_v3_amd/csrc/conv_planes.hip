// Implicit-GEMM convolution on the bf16 matrix cores over "bf16-plane" tensors.
//
// A tensor is stored as NP bf16 planes [NP][B,H,W,C]:
//   NP = 1  plain bf16 activations / weights (bf16 MFMA, fp32 accumulate)
//   NP = 3  an EXACT 3-way split of fp32 values, v = p0 + p1 + p2 (8+8+8 mantissa bits).  Every fp32
//           product is then evaluated as the six leading partial products
//               a0*b0 + a0*b1 + a1*b0 + a0*b2 + a1*b1 + a2*b0      (dropped terms <= 2^-26 |a*b|)
//           with fp32 accumulation: fp32-class error at 512/192 = 2.7x the fp32-MFMA rate
//           (six v_mfma_f32_32x32x16_bf16 @32 cycles instead of eight v_mfma_f32_32x32x2_f32 @64).
//
// Because both operands already live in memory as bf16 planes, tiles go HBM/L2 -> LDS by DMA
// (global_load_lds_dwordx4, 1 KiB per wave instruction) with no VGPR staging and no VALU work: a
// 3-stage LDS ring, counted s_waitcnt vmcnt(N) (never 0 in the loop) and ONE raw s_barrier per
// 16-deep K chunk.  The LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is
// applied to the per-lane SOURCE address and undone in the fragment reads (slot ^= (row>>3)&1 for
// 32-byte rows).  Halo / tail rows source a 64-byte zero page.  Weights are pre-arranged at pack time
// in exactly the LDS image order.  MFMA roles are swapped (A operand = weights, B operand = pixels)
// so each lane ends up with 4 consecutive output channels of one pixel: the epilogue (BN scale/shift,
// LeakyReLU, residual, re-split into planes) stores 8 bytes per plane per lane.
//
// Replaces the same reference call sites as conv_igemm_f32.hip (darknet.py:43-44, :52-53, :118,
// :161-162).
#include "yv3_common.h"

namespace {

__device__ __attribute__((aligned(64))) u16 g_zero_page[64];     // zero-initialised: source of halo rows

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
    int tb;                     // rows per packed weight tile
};

constexpr int PBK = 32;           // K elements per chunk
constexpr int ROWB = PBK * 2;     // bytes per tile row per plane (64: half a cache line)
constexpr int SLOTS = PBK / 8;    // 16-byte slots per row
constexpr int RPG = 64 / SLOTS;   // rows moved by one global_load_lds wave instruction (16)

typedef short bf16x8v __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ inline unsigned pack2_bf16_rn(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ inline float bf_lo(unsigned q) { return __uint_as_float(q << 16); }
__device__ inline float bf_hi(unsigned q) { return __uint_as_float(q & 0xffff0000u); }

// bank-conflict swizzle for 64-byte rows read with ds_read_b128: four rows share a 256-byte bank row
__device__ __host__ inline int swz(int row) { return (row >> 2) & (SLOTS - 1); }

template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int NP, int BM, int BN, int WM, int WN, int NSTAGE, bool K3, bool DUAL, bool OUT_F32>
__global__ __launch_bounds__(64 * WM * WN) void conv_planes_kernel(const ConvParamsP p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;          // wave tile: WTM pixels x WTN channels
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int A_PLANE = BM * ROWB;                    // bytes
    constexpr int B_PLANE = BN * ROWB;
    constexpr int STAGE = NP * (A_PLANE + B_PLANE);
    constexpr int AROWS = BM / NW;                        // pixel rows staged per wave (multiple of 16)
    constexpr int AQ = AROWS / RPG;                       // global_load_lds per plane per wave, A side
    constexpr int BROWS = BN / NW;                        // weight rows staged per wave (<= 16)
    constexpr int G = NP * (AQ + 1);                      // DMA instructions per chunk per wave
    constexpr int D = NSTAGE - 1;                         // prefetch distance in chunks
    static_assert(AROWS % RPG == 0 && BROWS <= RPG && MT >= 1 && NT >= 1, "tile/wave layout");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int bid = yv3_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % p.ntiles) * BN;
    const int m0 = (bid / p.ntiles) * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;

    // ---- staging, pixel side: wave w moves rows [AROWS*w, AROWS*(w+1)) of every plane, RPG rows per DMA.
    // lane -> (row = lane/4 within the group, physical slot = lane%4); the source is the un-swizzled slot.
    const int sslot = ((lane & (SLOTS - 1)) ^ ((lane >> 4) & (SLOTS - 1))) * 8;
    long long aoff[AQ], aoff2[DUAL ? AQ : 1];
    int ahi[K3 ? AQ : 1], awi[K3 ? AQ : 1];
    bool aok[AQ];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        const int m = m0 + AROWS * wid + q * RPG + (lane >> 2);
        aok[q] = m < p.M;
        const int mm = aok[q] ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        if (K3) {
            ahi[q] = ho * p.stride - 1; awi[q] = wo * p.stride - 1;
            aoff[q] = (((long long)b * p.H + ahi[q]) * p.W + awi[q]) * p.Cin + sslot;
        } else if (DUAL) {
            aoff[q] = (((long long)b * (p.H >> 1) + (ho >> 1)) * (p.W >> 1) + (wo >> 1)) * p.Cup + sslot;
            aoff2[q] = (((long long)b * p.H + ho) * p.W + wo) * (p.Cin - p.Cup) + sslot;
        } else {
            aoff[q] = (((long long)b * p.H + ho * p.stride) * p.W + wo * p.stride) * p.Cin + sslot;
        }
    }
    // ---- staging, weight side: the packed tile is already in LDS-image (swizzled) order
    const bool bact = lane < BROWS * SLOTS;
    const int brow = n0 % p.tb + BROWS * wid + (lane >> 2);
    const long long btile = (long long)(n0 / p.tb) * p.nk;
    const int bin = brow * PBK + (lane & (SLOTS - 1)) * 8;

    int kh = 0, kw = 0, c0 = 0;

    // ---- DMA of one K chunk = G wave instructions ("pieces").  dma_prepare computes this lane's source
    // pointers once per chunk; dma_piece(idx) issues one global_load_lds, so that the pieces can be spread
    // between the MFMAs of the previous chunk instead of being issued as one burst behind the barrier.
    const u16* ap[AQ];
    long long aps[AQ];
    const u16* wbp = p.w;
    unsigned char* dst = lds;
    auto dma_prepare = [&](int kc, int stage) {
        dst = lds + stage * STAGE;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            bool ok = aok[q];
            const u16* src = p.x;
            long long off, ps = p.xs;
            if (K3) {
                ok = ok && (unsigned)(ahi[q] + kh) < (unsigned)p.H && (unsigned)(awi[q] + kw) < (unsigned)p.W;
                off = aoff[q] + ((long long)kh * p.W + kw) * p.Cin + c0;
            } else if (DUAL) {
                if (c0 < p.Cup) off = aoff[q] + c0;
                else { src = p.x2; off = aoff2[q] + (c0 - p.Cup); ps = p.x2s; }
            } else {
                off = aoff[q] + c0;
            }
            ap[q] = ok ? src + off : g_zero_page;
            aps[q] = ok ? ps : 0;
        }
        wbp = p.w + ((btile + kc) * NP) * (long long)(p.tb * PBK) + bin;
        c0 += PBK;
        if (c0 == p.Cin) { c0 = 0; if (++kw == 3) { kw = 0; ++kh; } }
    };
    auto dma_piece = [&](int idx) {
        if (idx < AQ * NP) {
            const int q = idx / NP, pl = idx % NP;
            __builtin_amdgcn_global_load_lds(GPTR(ap[q] + pl * aps[q]),
                                             LPTR(dst + pl * A_PLANE + (AROWS * wid + q * RPG) * ROWB), 16, 0, 0);
        } else if (bact) {
            const int pl = idx - AQ * NP;
            __builtin_amdgcn_global_load_lds(GPTR(wbp + (long long)pl * (p.tb * PBK)),
                                             LPTR(dst + NP * A_PLANE + pl * B_PLANE + wid * (BROWS * ROWB)), 16, 0, 0);
        }
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    const int x_row = (wm * WTM + l31) * ROWB;                                  // pixel fragments (B operand)
    const int w_row = NP * A_PLANE + (wn * WTN + l31) * ROWB;                   // weight fragments (A operand)
    const int fsw = swz(l31);

#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < p.nk) {
            dma_prepare(d, d);
#pragma unroll
            for (int g = 0; g < G; ++g) dma_piece(g);
        }

    constexpr int KS = PBK / 16;              // MFMA k-steps per chunk
    constexpr int NF = (NT + MT) * NP;        // fragments (ds_read_b128) per k-step
    constexpr int NU = NT * MT;               // MFMA units (6 or 1 MFMAs each) per k-step
    bf16x8v frag[KS][NF];                     // [0, NT*NP): weights (i, plane); then pixels (j, plane)
    const unsigned char* st = lds;
    auto read_frag = [&](int ks, int f) {
        const int fslot = ((ks * 2 + lhi) ^ fsw) * 16;
        if (f < NT * NP) frag[ks][f] = *reinterpret_cast<const bf16x8v*>(st + w_row + fslot + (f / NP) * 32 * ROWB + (f % NP) * B_PLANE);
        else { const int g = f - NT * NP;
               frag[ks][f] = *reinterpret_cast<const bf16x8v*>(st + x_row + fslot + (g / NP) * 32 * ROWB + (g % NP) * A_PLANE); }
    };

    int cur = 0, nxt = D % NSTAGE;
    for (int kc = 0; kc < p.nk; ++kc) {
        // chunk kc must have landed; up to D-1 younger chunks may stay in flight (never a full drain mid-loop)
        if (kc + D - 1 < p.nk) wait_vmcnt<(D - 1) * G>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        st = lds + cur * STAGE;
#if !defined(YV3_ABLATE) || (YV3_ABLATE != 1)
        const bool more = kc + D < p.nk;
#else
        const bool more = false;
#endif
#if !defined(YV3_ABLATE) || (YV3_ABLATE != 2)
#pragma unroll
        for (int f = 0; f < NF; ++f) read_frag(0, f);
#endif
        if (more) dma_prepare(kc + D, nxt);
        __builtin_amdgcn_sched_barrier(0);
        // Units of MFMAs with the remaining LDS reads and the DMA pieces of chunk kc+D spread between
        // them: every non-MFMA instruction issues while the matrix pipe is busy with the previous unit.
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int gu = ks * NU + u;                              // unit index within the chunk
#if !defined(YV3_ABLATE) || (YV3_ABLATE != 2)
                if (ks + 1 < KS) {
#pragma unroll
                    for (int f = u * NF / NU; f < (u + 1) * NF / NU; ++f) read_frag(ks + 1, f);
                }
#endif
                if (more) {
#pragma unroll
                    for (int g = gu * G / (KS * NU); g < (gu + 1) * G / (KS * NU); ++g) dma_piece(g);
                }
#if !defined(YV3_ABLATE) || (YV3_ABLATE != 2)
                const int i = u / MT, j = u % MT;
                f32x16 c = acc[i][j];
                const bf16x8v* wf = &frag[ks][i * NP];
                const bf16x8v* xf = &frag[ks][NT * NP + j * NP];
                if constexpr (NP == 3) {
                    // smallest partial products first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2], xf[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], xf[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], xf[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[1], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[0], c, 0, 0, 0);
                acc[i][j] = c;
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
        nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }

    // ---- epilogue.  D tile of an MFMA: col = lane&31 -> pixel, row = (e&3) + 8*(e>>2) + 4*(lane>>5) -> channel.
    // A lane therefore holds 4 consecutive channels of ONE pixel; storing that directly scatters 8-byte
    // pieces over 32 rows per instruction.  Instead each wave transposes its WTM x WTN tile through LDS
    // (the pipeline stages are free now) so that 8 (or 4) neighbouring lanes cover the contiguous channels
    // of one pixel: residual planes are read and output planes written as full 16-byte-per-lane rows.
    constexpr int EP = WTN + 4;                       // floats per tile row (+4: conflict-free ds_write_b128)
    // (launch_cfg sizes the dynamic LDS as max(pipeline, NW * WTM * EP * 4))
    __syncthreads();                                  // every wave is done with the last stage
    float* tile = reinterpret_cast<float*>(lds) + wid * (WTM * EP);
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = i * 32 + 8 * g + 4 * lhi;                 // channel inside the wave tile
                const int n = n0 + wn * WTN + nl;
                f32x4 al = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
                if (OUT_F32) {          // head conv: cout (255) is not a multiple of 4 -> element-wise
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < p.Cout) { be[q] = p.beta[n + q]; if (p.alpha) al[q] = p.alpha[n + q]; }
                } else if (n < p.Cout) {
                    be = *reinterpret_cast<const f32x4*>(p.beta + n);
                    if (p.alpha) al = *reinterpret_cast<const f32x4*>(p.alpha + n);
                }
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = fmaf(acc[i][j][4 * g + q], al[q], be[q]);
                    if (p.act == YV3_ACT_LEAKY) t = t > 0.f ? t : 0.1f * t;
                    v[q] = t;
                }
                *reinterpret_cast<f32x4*>(tile + (j * 32 + l31) * EP + nl) = v;
            }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int LPR = WTN / 8;                      // lanes per pixel row (8 channels each)
    constexpr int RPP = 64 / LPR;                     // pixel rows per pass
#pragma unroll
    for (int ps = 0; ps < WTM / RPP; ++ps) {
        const int r = ps * RPP + lane / LPR;
        const int cg = (lane % LPR) * 8;
        const int m = m0 + wm * WTM + r;
        const int n = n0 + wn * WTN + cg;
        if (m >= p.M || n >= p.Cout) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(tile + r * EP + cg);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(tile + r * EP + cg + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const long long o = (long long)m * p.Cout + n;
        if (OUT_F32) {
            float* yo = (float*)p.y + o;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (n + q < p.Cout) yo[q] = v[q];
        } else {
            if (p.res) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {       // planes sum back to the exact fp32 value
                    const u32x4 q4 = *reinterpret_cast<const u32x4*>(p.res + pl * p.ys + o);
#pragma unroll
                    for (int h = 0; h < 4; ++h) { v[2 * h] += bf_lo(q4[h]); v[2 * h + 1] += bf_hi(q4[h]); }
                }
            }
            u16* yo = (u16*)p.y + o;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                u32x4 q4;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    q4[h] = pack2_bf16_rn(v[2 * h], v[2 * h + 1]);
                    v[2 * h] -= bf_lo(q4[h]); v[2 * h + 1] -= bf_hi(q4[h]);
                }
                *reinterpret_cast<u32x4*>(yo + pl * p.ys) = q4;
            }
        }
    }
}

template <int NP, int BM, int BN, int WM, int WN, int NSTAGE>
int launch_cfg(const ConvParamsP& p, bool k3, bool dual, bool out_f32, hipStream_t s) {
    const int mtiles = (p.M + BM - 1) / BM;
    const dim3 grid((unsigned)(mtiles * p.ntiles));
    const dim3 block(64 * WM * WN);
    const size_t pipe = (size_t)NSTAGE * NP * (BM + BN) * ROWB;
    const size_t epi = (size_t)WN * BM * (BN / WN + 4) * 4;   // WM*WN waves x (BM/WM) rows x (BN/WN + 4) floats
    const size_t lds = pipe > epi ? pipe : epi;
#define YV3_LAUNCH(K3_, DUAL_, OF_) \
    hipLaunchKernelGGL((conv_planes_kernel<NP, BM, BN, WM, WN, NSTAGE, K3_, DUAL_, OF_>), grid, block, lds, s, p)
    if (out_f32) {
        if (k3 || dual) return YV3_ESHAPE;                   // fp32 outputs are the 1x1 head convs
        YV3_LAUNCH(false, false, true);
    } else if (k3) YV3_LAUNCH(true, false, false);
    else if (dual) YV3_LAUNCH(false, true, false);
    else YV3_LAUNCH(false, false, false);
#undef YV3_LAUNCH
    YV3_CHECK_LAUNCH();
    return 0;
}

// OIHW fp32 -> NP bf16 planes in LDS-image order:
//   [n / tb][k / PBK][plane][n % tb][physical slot][8],  physical slot = (k % PBK)/8 ^ swz(n % tb),
//   k = (kh*3+kw)*cin + c
template <int NP>
__global__ void pack_weight_planes_kernel(const float* __restrict__ in, u16* __restrict__ out,
                                          int cout, int cin, int k, int tb, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // over [cout_pad][K]
    if (i >= total) return;
    const int kk = k * k;
    const int K = kk * cin;
    const int n = (int)(i / K);
    const int kidx = (int)(i - (long long)n * K);
    const int tap = kidx / cin, c = kidx - tap * cin;
    float v = 0.f;
    if (n < cout) v = in[((long long)n * cin + c) * kk + tap];
    const int nk = K / PBK;
    const int r = n % tb, ke = kidx % PBK;
    const int slot = (ke >> 3) ^ swz(r);
    const long long base = (((long long)(n / tb) * nk + kidx / PBK) * NP) * (long long)(tb * PBK) + r * PBK + slot * 8 + (ke & 7);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const u16 h = yv3_f2bf(v);
        out[base + (long long)pl * tb * PBK] = h;
        v -= yv3_bf2f(h);
    }
}

// fp32 [n] -> NP planes (RN split); used for layout conversion of whole tensors
template <int NP>
__global__ void split_planes_kernel(const float* __restrict__ in, u16* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = in[i];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const u16 h = yv3_f2bf(v);
        out[i + pl * n] = h;
        v -= yv3_bf2f(h);
    }
}

template <int NP>
__global__ void merge_planes_kernel(const u16* __restrict__ in, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) v += yv3_bf2f(in[i + pl * n]);
    out[i] = v;
}

}  // namespace

int yv3_pack_weight_planes(const float* w_oihw, void* w_packed, int cout, int cin, int k, int cout_pad, int np, hipStream_t s) {
    const long long total = (long long)cout_pad * k * k * cin;
    const int tb = cout_pad < 128 ? cout_pad : 128;
    if (cout_pad % tb) return YV3_ESHAPE;
    const dim3 grid(yv3_ceil_div(total, 256));
    if (np == 3) hipLaunchKernelGGL(pack_weight_planes_kernel<3>, grid, dim3(256), 0, s, w_oihw, (u16*)w_packed, cout, cin, k, tb, total);
    else         hipLaunchKernelGGL(pack_weight_planes_kernel<1>, grid, dim3(256), 0, s, w_oihw, (u16*)w_packed, cout, cin, k, tb, total);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_split_planes(const float* in, void* out, long long n, int np, void* stream) {
    if (!in || !out || n < 0 || (np != 1 && np != 3)) return YV3_EINVAL;
    if (n == 0) return 0;
    const dim3 grid(yv3_ceil_div(n, 256));
    if (np == 3) hipLaunchKernelGGL(split_planes_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, in, (u16*)out, n);
    else         hipLaunchKernelGGL(split_planes_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, in, (u16*)out, n);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_merge_planes(const void* in, float* out, long long n, int np, void* stream) {
    if (!in || !out || n < 0 || (np != 1 && np != 3)) return YV3_EINVAL;
    if (n == 0) return 0;
    const dim3 grid(yv3_ceil_div(n, 256));
    if (np == 3) hipLaunchKernelGGL(merge_planes_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)in, out, n);
    else         hipLaunchKernelGGL(merge_planes_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)in, out, n);
    YV3_CHECK_LAUNCH();
    return 0;
}

int yv3_conv2d_planes(const yv3_conv_desc* d, int np, hipStream_t s) {
    ConvParamsP p;
    p.x = (const u16*)d->x; p.x2 = (const u16*)d->x2; p.w = (const u16*)d->w;
    p.alpha = d->alpha; p.beta = d->beta; p.res = (const u16*)d->residual; p.y = d->y;
    p.H = d->H; p.W = d->W; p.Cin = d->cin; p.Cup = d->cin_up; p.Cout = d->cout;
    p.stride = d->stride; p.act = d->act;
    const int pad = (d->k - 1) / 2;
    p.Ho = (d->H + 2 * pad - d->k) / d->stride + 1;
    p.Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
    const long long M = (long long)d->B * p.Ho * p.Wo;
    if (M > 0x7fffffffLL) return YV3_ESHAPE;
    p.M = (int)M;
    p.K = d->k * d->k * d->cin;
    p.nk = p.K / PBK;
    if (d->cin_up) {
        p.xs = (long long)d->B * (d->H / 2) * (d->W / 2) * d->cin_up;
        p.x2s = (long long)d->B * d->H * d->W * (d->cin - d->cin_up);
    } else {
        p.xs = (long long)d->B * d->H * d->W * d->cin;
        p.x2s = 0;
    }
    p.ys = M * d->cout;
    const bool out_f32 = d->out_dtype == YV3_F32;
    if (!out_f32 && (d->cout % 8)) return YV3_ESHAPE;
    const int npad = d->cout_pad;
    p.tb = npad < 128 ? npad : 128;
    if (npad % p.tb) return YV3_ESHAPE;
    const bool k3 = d->k == 3, dual = d->cin_up > 0;
#define YV3_CFG(BM_, BN_, WM_, WN_, NS_) (np == 3 ? launch_cfg<3, BM_, BN_, WM_, WN_, NS_>(p, k3, dual, out_f32, s) \
                                                   : launch_cfg<1, BM_, BN_, WM_, WN_, NS_>(p, k3, dual, out_f32, s))
    if (npad % 128 == 0) {
        // 256x128 tiles (8 waves, 144 KB LDS) when they still give every CU >= 2 rounds of work,
        // else 128x128 tiles (8 waves of 32x64) for finer granularity on the 13x13 / 26x26 layers
        const long long blocks256 = ((M + 255) / 256) * (npad / 128);
        p.ntiles = npad / 128;
        if (blocks256 >= 512) return YV3_CFG(256, 128, 4, 2, 2);
        return YV3_CFG(128, 128, 4, 2, 3);
    }
    if (npad % 64 == 0) { p.ntiles = npad / 64; return YV3_CFG(128, 64, 2, 2, 2); }
    p.ntiles = npad / 32;
    return YV3_CFG(128, 32, 4, 1, 2);
#undef YV3_CFG
}
