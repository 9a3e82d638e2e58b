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
#include <stdlib.h>
#include "conv_planes_common.h"

namespace {

#ifndef YV3_PP_DMA
#define YV3_PP_DMA 1
#endif
#ifndef YV3_PP_GRP
#define YV3_PP_GRP(wid) ((wid) >> 2)
#endif

// PP ("ping-pong"): the 8 waves of the workgroup form two groups of four (one wave per SIMD each) that run
// half a chunk out of phase: while one group issues a chunk's 24 MFMAs from registers, the other reads its
// fragments of the next chunk from LDS and issues its share of the DMA, then they swap at an s_barrier.
// Each SIMD's matrix pipe is thereby fed by one wave while its partner wave loads, instead of both
// waves stalling on LDS / DMA / barrier at the same time.
template <int NP, int BM, int BN, int WM, int WN, int NSTAGE, bool K3, bool DUAL, bool OUT_F32, bool PP = false>
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
#ifdef YV3_TIMELINE
    const unsigned long long tl_entry = __builtin_amdgcn_s_memtime();
#endif

    if (PP && p.stagger && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }
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
    // The per-lane source pointer only changes shape when the filter tap changes (every Cin/32 chunks): bounds
    // test, base pointer and plane stride are recomputed there; inside a tap a chunk is one 64-bit add.
    const u16* ap[AQ];
    long long aps[AQ];
    int ainc[AQ];
    const u16* wbp = p.w;
    unsigned char* dst = lds;
    auto dma_prepare = [&](int kc, int stage) {
        dst = lds + stage * STAGE;
        if (DUAL) {
#pragma unroll
            for (int q = 0; q < AQ; ++q) {
                const bool ok = aok[q];
                const u16* src = p.x;
                long long off, ps = p.xs;
                if (c0 < p.Cup) off = aoff[q] + c0;
                else { src = p.x2; off = aoff2[q] + (c0 - p.Cup); ps = p.x2s; }
                ap[q] = ok ? src + off : g_zero_page;
                aps[q] = ok ? ps : 0;
            }
        } else if (c0 == 0) {                                  // wave-uniform: first chunk of a tap (or of a 1x1 conv)
#pragma unroll
            for (int q = 0; q < AQ; ++q) {
                bool ok = aok[q];
                long long off = aoff[q];
                if (K3) {
                    ok = ok && (unsigned)(ahi[q] + kh) < (unsigned)p.H && (unsigned)(awi[q] + kw) < (unsigned)p.W;
                    off += ((long long)kh * p.W + kw) * p.Cin;
                }
                ap[q] = ok ? p.x + off : g_zero_page;
                aps[q] = ok ? p.xs : 0;
                ainc[q] = ok ? PBK : 0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < AQ; ++q) ap[q] += ainc[q];
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

    if constexpr (PP) {
        static_assert(!PP || (NW == 8 && NSTAGE >= 3 && NP == 2), "ping-pong: 8 waves, 3-deep ring, fp16x2 planes");
        const int grp = YV3_PP_GRP(wid);
        // 32x64 wave tiles (128x128 workgroup tile): the compute segment is only 12 MFMAs per k-step, so the second
        // k-step's fragments are fetched under the first k-step's MFMAs; measured +5 % there, -5 % on 64x64 wave tiles
        constexpr bool SPLIT = MT == 1;
        if (D <= p.nk) wait_vmcnt<(D - 1) * G>(); else wait_vmcnt<0>();       // chunk 0 has landed
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();                           // group 1 runs one segment behind
#ifdef YV3_TIMELINE
        const unsigned long long tl_loop0 = __builtin_amdgcn_s_memtime();
        unsigned long long tl_load = 0, tl_b1 = 0, tl_comp = 0, tl_b2 = 0, tl_t = tl_loop0;
        unsigned long long tl_l1 = 0, tl_l2 = 0, tl_l3 = 0;
#define TL_MARK(acc_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - tl_t; tl_t = t_; } while (0)
#else
#define TL_MARK(acc_) do {} while (0)
#endif
        int cur = 0, nxt = D % NSTAGE;
        for (int kc = 0; kc < p.nk; ++kc) {
            // ---- load segment: fragments of chunk kc -> registers, DMA of chunk kc+D -> the stage chunk kc-1 used
            st = lds + cur * STAGE;
#pragma unroll
            for (int ks = 0; ks < (SPLIT ? 1 : KS); ++ks)
#pragma unroll
                for (int f = 0; f < NF; ++f) read_frag(ks, f);
            const bool more = kc + D < p.nk;
#ifdef YV3_TIMELINE
            __builtin_amdgcn_sched_barrier(0);
            { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_l1 += t_ - tl_t; }
#endif
#if YV3_PP_DMA == 0
            if (more) {
                dma_prepare(kc + D, nxt);
#pragma unroll
                for (int g = 0; g < G; ++g) dma_piece(g);
                wait_vmcnt<(D - 1) * G>();                                    // my pieces of chunk kc+1 have landed
            } else {
                wait_vmcnt<0>();
            }
#else
            // addresses now (VALU, off the matrix pipe's critical path); the pieces go out between the MFMAs of
            // the compute segment.  My pieces of chunk kc+1 (issued D-1 compute segments ago) must have landed.
            if (more) dma_prepare(kc + D, nxt);
#ifdef YV3_TIMELINE
            __builtin_amdgcn_sched_barrier(0);
            { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_l2 += t_ - tl_t; }
#endif
            if (kc + D - 1 < p.nk) wait_vmcnt<(D - 2) * G>(); else wait_vmcnt<0>();
#ifdef YV3_TIMELINE
            { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_l3 += t_ - tl_t; }
#endif
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            TL_MARK(tl_load);
            __builtin_amdgcn_s_barrier();
            TL_MARK(tl_b1);
            __builtin_amdgcn_sched_barrier(0);
            // ---- compute segment: registers only (SPLIT: plus the second k-step's fragment reads)
            if constexpr (SPLIT) {
#pragma unroll
                for (int ks = 1; ks < KS; ++ks)
#pragma unroll
                    for (int f = 0; f < NF; ++f) read_frag(ks, f);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (SPLIT && ks == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (NP == 2) {                                       // rotate over the accumulators: no back-to-back RAW
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            const int i = u / MT, j = u % MT;
                            const bf16x8v* wf = &frag[ks][i * NP];
                            const bf16x8v* xf = &frag[ks][NT * NP + j * NP];
                            acc[i][j] = PlaneOps<2>::mfma(wf[t == 0 ? 1 : 0], xf[t == 1 ? 1 : 0], acc[i][j]);
#if YV3_PP_DMA != 0
                            {   // one DMA piece after every (24 / G)-th MFMA
                                constexpr int TOT = KS * 3 * NU;
                                const int mi = (ks * 3 + t) * NU + u;
                                if (more && (mi * G) / TOT != ((mi + 1) * G) / TOT) dma_piece((mi * G) / TOT);
                            }
#endif
                        }
                } else {
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int i = u / MT, j = u % MT;
                        acc[i][j] = mfma_unit<NP>(&frag[ks][i * NP], &frag[ks][NT * NP + j * NP], acc[i][j]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            TL_MARK(tl_comp);
            if (!(grp == 1 && kc + 1 == p.nk)) __builtin_amdgcn_s_barrier();  // both groups pass 2*nk + 1 barriers
            TL_MARK(tl_b2);
            __builtin_amdgcn_sched_barrier(0);
            cur = cur + 1 == NSTAGE ? 0 : cur + 1;
            nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
        }
        // group 0 starts its epilogue while group 1 still computes (group 1 reads no LDS in its last segment and
        // the epilogue's LDS tiles are per wave)
#ifdef YV3_TIMELINE
        const unsigned long long tl_loop1 = __builtin_amdgcn_s_memtime();
#endif
        epilogue_store<NP, BM, BN, WM, WN, OUT_F32, false>(acc, p, lds, m0, n0, wid, lane);
#ifdef YV3_TIMELINE
        const unsigned long long tl_epi = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tl_drain = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 300 && lane == 0 && p.alpha) {     // debug build only: cycle split of one block -> alpha[0..]
            float* dbg = const_cast<float*>(p.alpha) + wid * 8;
            dbg[0] = (float)(tl_loop0 - tl_entry); dbg[1] = (float)tl_load; dbg[2] = (float)tl_b1; dbg[3] = (float)tl_comp;
            dbg[4] = (float)tl_b2; dbg[5] = (float)(tl_epi - tl_loop1); dbg[6] = (float)(tl_drain - tl_epi); dbg[7] = (float)(tl_drain - tl_entry);
            dbg[64] = (float)tl_l1; dbg[65] = (float)tl_l2; dbg[66] = (float)tl_l3;
        }
#endif
        return;
    }
#ifdef YV3_TIMELINE
    unsigned long long tl_wait = 0, tl_bar = 0, tl_body = 0, tl_prev = 0, tl_dma = 0;
#endif
#if defined(YV3_PRIO) && YV3_PRIO == 1
    // the second-dispatched half of an 8-wave workgroup loses issue arbitration (age) to the first half on
    // every segment and the first half then idles at the barrier; static priority for the younger half
    if (NW == 8 && wid >= 4) __builtin_amdgcn_s_setprio(1);
#elif defined(YV3_PRIO) && YV3_PRIO == 2
    if (NW == 8 && wid < 4) __builtin_amdgcn_s_setprio(1);
#endif
    int cur = 0, nxt = D % NSTAGE;
    for (int kc = 0; kc < p.nk; ++kc) {
        // chunk kc must have landed; up to D-1 younger chunks may stay in flight (never a full drain mid-loop)
#ifdef YV3_TIMELINE
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        if (kc + D - 1 < p.nk) wait_vmcnt<(D - 1) * G>(); else wait_vmcnt<0>();
#ifdef YV3_TIMELINE
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef YV3_TIMELINE
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        tl_wait += t1 - t0; tl_bar += t2 - t1;
        if (kc > 0) tl_body += t0 - tl_prev;
        tl_prev = t2;
#endif
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
                    // all pieces go out during the FIRST k-step's units: the DMA then has the rest of this chunk's
                    // MFMAs to land before the wait at the top of the next iteration
                    constexpr int DU = YV3_DMA_UNITS < KS * NU ? YV3_DMA_UNITS : KS * NU;
#pragma unroll
                    for (int g = (gu < DU ? gu * G / DU : G); g < (gu < DU ? (gu + 1) * G / DU : G); ++g) {
#ifdef YV3_TIMELINE
                        const unsigned long long ta = __builtin_amdgcn_s_memtime();
                        dma_piece(g);
                        tl_dma += __builtin_amdgcn_s_memtime() - ta;
#else
                        dma_piece(g);
#endif
                    }
                }
#if defined(YV3_ABLATE) && (YV3_ABLATE == 3)
                {   // keep the fragment reads alive, skip the MFMAs: DMA + LDS-read path only
                    const int i = u / MT, j = u % MT;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        asm volatile("" :: "v"(frag[ks][i * NP + pl]));
                        asm volatile("" :: "v"(frag[ks][NT * NP + j * NP + pl]));
                    }
                }
#elif !defined(YV3_ABLATE) || (YV3_ABLATE != 2)
                const int i = u / MT, j = u % MT;
                f32x16 c = acc[i][j];
                const bf16x8v* wf = &frag[ks][i * NP];
                const bf16x8v* xf = &frag[ks][NT * NP + j * NP];
                c = mfma_unit<NP>(wf, xf, c);
                acc[i][j] = c;
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
        nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }

#ifdef YV3_TIMELINE
    if (blockIdx.x == 17 && lane == 0 && p.alpha) {      // debug build only: dump cycle split of one block into alpha[0..]
        float* dbg = const_cast<float*>(p.alpha);
        dbg[wid * 4 + 0] = (float)tl_wait; dbg[wid * 4 + 1] = (float)tl_bar; dbg[wid * 4 + 2] = (float)tl_body; dbg[wid * 4 + 3] = (float)tl_dma;
    }
#endif
    epilogue_store<NP, BM, BN, WM, WN, OUT_F32>(acc, p, lds, m0, n0, wid, lane);
}

template <int NP, int BM, int BN, int WM, int WN, int NSTAGE>
int launch_cfg(const ConvParamsP& p, bool k3, bool dual, bool out_f32, hipStream_t s) {
    const int mtiles = (p.M + BM - 1) / BM;
    const dim3 grid((unsigned)(mtiles * p.ntiles));
    const dim3 block(64 * WM * WN);
    const size_t pipe = (size_t)NSTAGE * NP * (BM + BN) * ROWB;
    const size_t epi = (size_t)WN * BM * (BN / WN + 4) * 4;   // WM*WN waves x (BM/WM) rows x (BN/WN + 4) floats
    const size_t lds = pipe > epi ? pipe : epi;
    static const bool use_pp = getenv("YV3_NO_PP") == nullptr;     // ping-pong main loop (fp16x2, 8-wave tiles) unless disabled
#define YV3_LAUNCH(K3_, DUAL_, OF_) do { \
    if constexpr (NP == 2 && WM * WN == 8 && NSTAGE >= 3) { \
        if (use_pp) { hipLaunchKernelGGL((conv_planes_kernel<NP, BM, BN, WM, WN, NSTAGE, K3_, DUAL_, OF_, true>), grid, block, lds, s, p); break; } \
    } \
    hipLaunchKernelGGL((conv_planes_kernel<NP, BM, BN, WM, WN, NSTAGE, K3_, DUAL_, OF_>), grid, block, lds, s, p); } while (0)
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
        const u16 h = PlaneOps<NP>::cvt(v);
        out[base + (long long)pl * tb * PBK] = h;
        v -= PlaneOps<NP>::back(h);
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
        const u16 h = PlaneOps<NP>::cvt(v);
        out[i + pl * n] = h;
        v -= PlaneOps<NP>::back(h);
    }
}

template <int NP>
__global__ void merge_planes_kernel(const u16* __restrict__ in, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) v += PlaneOps<NP>::back(in[i + pl * n]);
    out[i] = v;
}

}  // namespace

int yv3_pack_weight_planes(const float* w_oihw, void* w_packed, int cout, int cin, int k, int cout_pad, int np, hipStream_t s) {
    const long long total = (long long)cout_pad * k * k * cin;
    const int tb = cout_pad < 128 ? cout_pad : 128;
    if (cout_pad % tb) return YV3_ESHAPE;
    const dim3 grid(yv3_ceil_div(total, 256));
    if (np == 3)      hipLaunchKernelGGL(pack_weight_planes_kernel<3>, grid, dim3(256), 0, s, w_oihw, (u16*)w_packed, cout, cin, k, tb, total);
    else if (np == 2) hipLaunchKernelGGL(pack_weight_planes_kernel<2>, grid, dim3(256), 0, s, w_oihw, (u16*)w_packed, cout, cin, k, tb, total);
    else              hipLaunchKernelGGL(pack_weight_planes_kernel<1>, grid, dim3(256), 0, s, w_oihw, (u16*)w_packed, cout, cin, k, tb, total);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_split_planes(const float* in, void* out, long long n, int np, void* stream) {
    if (!in || !out || n < 0 || np < 1 || np > 3) return YV3_EINVAL;
    if (n == 0) return 0;
    const dim3 grid(yv3_ceil_div(n, 256));
    if (np == 3)      hipLaunchKernelGGL(split_planes_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, in, (u16*)out, n);
    else if (np == 2) hipLaunchKernelGGL(split_planes_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, in, (u16*)out, n);
    else              hipLaunchKernelGGL(split_planes_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, in, (u16*)out, n);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_merge_planes(const void* in, float* out, long long n, int np, void* stream) {
    if (!in || !out || n < 0 || np < 1 || np > 3) return YV3_EINVAL;
    if (n == 0) return 0;
    const dim3 grid(yv3_ceil_div(n, 256));
    if (np == 3)      hipLaunchKernelGGL(merge_planes_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)in, out, n);
    else if (np == 2) hipLaunchKernelGGL(merge_planes_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)in, out, n);
    else              hipLaunchKernelGGL(merge_planes_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)in, out, n);
    YV3_CHECK_LAUNCH();
    return 0;
}

int yv3_conv2d_planes_k3s1(const ConvParamsP* pp, int np, int npad, long long M, hipStream_t s);

int yv3_conv2d_planes(const yv3_conv_desc* d, int np, hipStream_t s) {
    ConvParamsP p;
    p.x = (const u16*)d->x; p.x2 = (const u16*)d->x2; p.w = (const u16*)d->w;
    p.alpha = d->alpha; p.beta = d->beta; p.res = (const u16*)d->residual; p.y = d->y;
    p.H = d->H; p.W = d->W; p.Cin = d->cin; p.Cup = d->cin_up; p.Cout = d->cout;
    p.stride = d->stride; p.act = d->act; p.flags = d->flags;
    { const char* e = getenv("YV3_STAGGER"); p.stagger = e ? atoi(e) : 0; }
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
    // kw-tap reuse kernel (conv_planes_k3s1.hip): 44 % less L2->LDS traffic, same results, but no faster on
    // MI355X because this MFMA stream is power-limited -- opt-in until that changes.
    if (k3 && d->stride == 1 && !out_f32 && getenv("YV3_K3S1")) {
        const int rc = yv3_conv2d_planes_k3s1(&p, np, npad, M, s);
        if (rc != -100) return rc;
    }
#define YV3_CFG(BM_, BN_, WM_, WN_, NS_) (np == 3 ? launch_cfg<3, BM_, BN_, WM_, WN_, NS_>(p, k3, dual, out_f32, s) : \
                                         np == 2 ? launch_cfg<2, BM_, BN_, WM_, WN_, (NS_) + 1>(p, k3, dual, out_f32, s) : \
                                                   launch_cfg<1, BM_, BN_, WM_, WN_, NS_>(p, k3, dual, out_f32, s))
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
