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
#include <type_traits>
#include "conv_planes_common.h"

namespace {

// s_waitcnt lgkmcnt(N) as the BUILTIN (vmcnt / expcnt left at their maxima): hipcc's own wait insertion understands it, so after it the
// compiler knows those LDS reads have returned.  Behind an inline-asm wait it does not -- and then puts a full lgkmcnt(0) in front of
// the next use of the registers, AFTER younger reads were issued (the second k-step's fragments requested at the start of a compute
// segment: ~320 exposed cycles per chunk, round 4).
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8)); }

#ifndef YV3_PP_GRP
#define YV3_PP_GRP(wid) ((wid) >> 2)
#endif
// Every compile-time measurement switch below (timeline dumps that overwrite alpha[], ablations with INVALID results, schedule
// experiments) exists only in measurement builds: the shipped libyv3.so is compiled without any of them.
#if !defined(YV3_MEASURE) && (defined(YV3_TIMELINE) || defined(YV3_ABLATE) || defined(YV3_WABL) || defined(YV3_PPX) || defined(YV3_PRIO) || \
                              defined(YV3_EXP_BF16MFMA) || defined(YV3_AB_NO_TWO_LANES_RULE))
#error "measurement switches need -DYV3_MEASURE (make measure / tools/build_variant.sh); the shipped library has none"
#endif
// Measurement switches.  Run time (yv3_conv_desc.tune[], set by tools/ through YV3_TUNE=a,b,c,d; 0 = the shipped behaviour):
//   tune[1] bit 0  no two-workgroup tile for the short-K 1x1 layers      bit 1  Winograd stage: rolling instead of ping-pong main loop      bit 4  bf16: no 192-row variant of the 256x256 tile
//           bit 3  bf16: round-3 tile selection (no rolling loop, no 256x256 tile)
//   tune[2]        bf16: threshold (256x128 tiles) from which the four-wave tile is used
//   tune[3] (measurement builds, -DYV3_MEASURE, only) bit 0  epilogue without its stores   bit 1  without its residual loads   (results INVALID)
// Compile time (A/B builds through tools/build_variant.sh):
// YV3_WABL (timing ablations of the Winograd GEMM stage's ping-pong loop, results INVALID; tools/timeline.py --kernel wino):
//   1 no DMA pieces in the compute segment   2 both k-steps' fragments read in the load segment (no SPLIT)
//   4 no fold at the end of a position       8 no MFMAs (fragments kept alive)
//   16 every DMA source is the tile's FIRST chunk (L1/L2-hot lines: no memory-system bandwidth / latency in the loop)
//   32 pixel-side rows paired into full 128-byte lines (row r reads half (r & 1) of line r >> 1: the traffic of 128-byte rows)
#ifndef YV3_WABL
#define YV3_WABL 0
#endif
// YV3_PPX (schedule experiments of the ping-pong loop, results valid): 1 s_setprio(1) around the compute segment's MFMAs
//   2 Winograd stage: both k-steps' fragments in the load segment (no SPLIT)   4 the same for every 32x64-wave-tile kernel
#ifndef YV3_PPX
#define YV3_PPX 0
#endif
// YV3_WINO_EPI4: the Winograd tile's four outputs through epilogue_store_wino4 (1) or four epilogue_store passes (0; A/B builds)
#ifndef YV3_WINO_EPI4
#define YV3_WINO_EPI4 1
#endif

// PP ("ping-pong"): the 8 waves of the workgroup form two groups of four (one wave per SIMD each) that run
// half a chunk out of phase: while one group issues a chunk's 24 MFMAs from registers, the other reads its
// fragments of the next chunk from LDS and issues its share of the DMA, then they swap at an s_barrier.
// Each SIMD's matrix pipe is thereby fed by one wave while its partner wave loads, instead of both
// waves stalling on LDS / DMA / barrier at the same time.
// MINW = 2 (4-wave workgroups only): two workgroups per CU (<= 256 registers per wave); their barriers are independent, so
// one workgroup's waves feed the matrix pipes while the other's wait for DMA / run their epilogue.
// MTG (0 = whole wave tile): 32-row blocks per epilogue round (conv_planes_common.h) -- 128-row wave tiles use 2.
// WINO (ping-pong kernels, NP = 2): Winograd F(2x2,3x3) GEMM stage (csrc/winograd.hip): the K loop walks the 16 transform
// positions (Cin/32 chunks each, operand matrix xi * xi_stride into the V planes); at the end of a position the product
// accumulators are folded into the tile's four outputs with the coefficients of A^T x A^T (0 / +-1: exact) and cleared.
// ROLL (single-phase kernels, ring of >= 3 stages): the chunk's ONE barrier sits between its two k-steps, and the first k-step's
// fragments of chunk k+1 are read under the second k-step's MFMAs of chunk k -- a rolling software pipeline over the chunk
// boundary, so that no LDS read latency is ever exposed behind a barrier (the plain loop reads a chunk's first fragments right
// after its barrier and waits for them before the first MFMA).
template <int NP, int BM, int BN, int WM, int WN, int NSTAGE, bool K3, bool DUAL, bool OUT_F32, bool PP = false, bool SK = false, int MINW = 1, int MTG = 0,
          bool WINO = false, bool ROLL = false>
__global__ __launch_bounds__(64 * WM * WN, MINW) void conv_planes_kernel(const ConvParamsP p) {
    static_assert(!WINO || ((PP || ROLL) && !K3 && !DUAL && !OUT_F32 && NP == 2), "Winograd stage: ping-pong or rolling fp16-plane kernel");
    static_assert(!ROLL || (!PP && !SK && NSTAGE >= 3), "rolling loop: single-phase kernel with a ring of >= 3 stages");
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;          // wave tile: WTM pixels x WTN channels
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int EMTG = MTG ? MTG : MT;
    constexpr int A_PLANE = BM * ROWB;                    // bytes
    constexpr int B_PLANE = BN * ROWB;
    constexpr int STAGE = NP * (A_PLANE + B_PLANE);
    constexpr int AROWS = BM / NW;                        // pixel rows staged per wave (multiple of 16)
    constexpr int AQ = (AROWS + RPG - 1) / RPG;           // global_load_lds per plane per wave, A side
    constexpr int ATAIL = AROWS % RPG;                    // rows of the last one when it is partial (192-row tiles on 8 waves: 24 = 16 + 8)
    constexpr int BROWS = BN / NW;                        // weight rows staged per wave
    constexpr int BQ = BROWS > RPG ? BROWS / RPG : 1;     // global_load_lds per plane per wave, weight side
    constexpr int G = NP * (AQ + BQ);                     // DMA instructions per chunk per wave
    constexpr int D = NSTAGE - 1;                         // prefetch distance in chunks
    static_assert((BROWS <= RPG || BROWS % RPG == 0) && MT >= 1 && NT >= 1, "tile/wave layout");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef YV3_TIMELINE
    const unsigned long long tl_entry = __builtin_amdgcn_s_memtime();
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;

    // ---- staging, pixel side: wave w moves rows [AROWS*w, AROWS*(w+1)) of every plane, RPG rows per DMA.
    // lane -> (row = lane/4 within the group, physical slot = lane%4); the source is the un-swizzled slot.
    // (the swizzle follows the row's position in the TILE: a wave's first row is a multiple of 16 except with a partial last piece)
    const int sslot = ((lane & (SLOTS - 1)) ^ (((lane >> 4) + (ATAIL ? (AROWS * wid) >> 2 : 0)) & (SLOTS - 1))) * 8;
    long long aoff[AQ], aoff2[DUAL ? AQ : 1];
    int ahi[K3 ? AQ : 1], awi[K3 ? AQ : 1];
    bool aok[AQ];
    const int HoWo = p.Ho * p.Wo;
    // ---- staging, weight side: the packed tile is already in LDS-image (swizzled) order
    const bool bact = lane < (BROWS < RPG ? BROWS : RPG) * SLOTS;
    int m0 = 0, n0 = 0, bin = 0;
    long long btile = 0;
    int kh = 0, kw = 0, c0 = 0;
    bool tapinit = true;
    // index math of one workgroup tile (logical id `bid`), positioned at K chunk k0
    auto setup_tile = [&](int bid, int k0) {
        n0 = (bid % p.ntiles) * BN;
        m0 = (bid / p.ntiles) * BM;
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
                if constexpr (WINO && (YV3_WABL & 32)) aoff[q] = (long long)(mm >> 1) * p.Cin + (mm & 1) * 32 + sslot;
            }
        }
        const int n0w = n0 + BROWS * wid;                      // first weight row of this wave (a wave's rows never straddle a packed tile)
        const int brow = n0w % p.tb + (lane >> 2);
        btile = (long long)(n0w / p.tb) * p.nk;
        bin = brow * PBK + (lane & (SLOTS - 1)) * 8;
        const int cpt = p.Cin / PBK;                           // chunks per filter tap
        const int tap = (K3 || WINO) ? k0 / cpt : 0;
        kh = WINO ? 0 : tap / 3; kw = tap - kh * 3; c0 = (k0 - tap * cpt) * PBK;      // (WINO: kw = transform position)
        tapinit = true;
    };
    if (!(PP && SK)) setup_tile(yv3_xcd_remap(blockIdx.x, gridDim.x), 0);

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
        } else if (c0 == 0 || tapinit) {                       // wave-uniform: first chunk of a tap / of the tile
            tapinit = false;
#pragma unroll
            for (int q = 0; q < AQ; ++q) {
                bool ok = aok[q];
                long long off = aoff[q] + c0;
                if (K3) {
                    ok = ok && (unsigned)(ahi[q] + kh) < (unsigned)p.H && (unsigned)(awi[q] + kw) < (unsigned)p.W;
                    off += ((long long)kh * p.W + kw) * p.Cin;
                }
                if (WINO) off += (long long)kw * p.xi_stride;
                ap[q] = ok ? p.x + off : g_zero_page;
                aps[q] = ok ? p.xs : 0;
                ainc[q] = ok ? PBK : 0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < AQ; ++q) ap[q] += ainc[q];
        }
        wbp = p.w + ((btile + kc) * NP) * (long long)(p.tb * PBK) + bin;
        if constexpr (WINO && (YV3_WABL & 16)) {
            wbp = p.w + (btile * NP) * (long long)(p.tb * PBK) + bin;
#pragma unroll
            for (int q = 0; q < AQ; ++q) { ap[q] = p.x + aoff[q]; aps[q] = p.xs; }
        }
        c0 += PBK;
        if (c0 == p.Cin) { c0 = 0; if (WINO) ++kw; else if (++kw == 3) { kw = 0; ++kh; } }
    };
    auto dma_piece = [&](int idx) {
        if (idx < AQ * NP) {
            const int q = idx / NP, pl = idx % NP;
            // (a partial last piece: the lanes of the rows beyond this wave's share stay out -- they would land in the next wave's rows)
            if (ATAIL == 0 || q < AQ - 1 || (lane >> 2) < ATAIL)
                __builtin_amdgcn_global_load_lds(GPTR(ap[q] + pl * aps[q]),
                                                 LPTR(dst + pl * A_PLANE + (AROWS * wid + q * RPG) * ROWB), 16, 0, 0);
        } else if (bact) {
            const int q = (idx - AQ * NP) / NP, pl = (idx - AQ * NP) % NP;
            __builtin_amdgcn_global_load_lds(GPTR(wbp + (long long)pl * (p.tb * PBK) + q * (RPG * PBK)),
                                             LPTR(dst + NP * A_PLANE + pl * B_PLANE + (wid * BROWS + q * RPG) * ROWB), 16, 0, 0);
        }
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x16 yac[WINO ? 4 : 1][NT][MT];          // WINO: the tile's four outputs Y[wi][wj] (index 2*wi + wj)
    if constexpr (WINO) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) yac[o][i][j][e] = 0.f;
    }
    int wleft = WINO ? p.Cin / PBK : 0, wxi = 0;   // chunks left in the current transform position, its index

    const int l31 = lane & 31, lhi = lane >> 5;
    const int x_row = (wm * WTM + l31) * ROWB;                                  // pixel fragments (B operand)
    const int w_row = NP * A_PLANE + (wn * WTN + l31) * ROWB;                   // weight fragments (A operand)
    const int fsw = swz(l31);

    if constexpr (!PP) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < p.nk) {
                dma_prepare(d, d);
#pragma unroll
                for (int g = 0; g < G; ++g) dma_piece(g);
            }
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
        static_assert(!PP || (NW == 8 && NSTAGE >= 3 && NP <= 2), "ping-pong: 8 waves, 3-deep ring, one or two planes");
        constexpr int NMF = NP == 2 ? 3 : 1;      // MFMAs per (weight tile, pixel tile, k-step)
        const int grp = YV3_PP_GRP(wid);
        // 32x64 wave tiles (128x128 workgroup tile): the compute segment is only 12 MFMAs per k-step, so the second
        // k-step's fragments are fetched under the first k-step's MFMAs; measured +5 % there, -5 % on 64x64 wave tiles
        // (round 4: NOT for the Winograd stage -- with its 4-deep ring both k-steps' fragments in the load segment are 2-4 % faster per
        // layer, profiles/r04c_pp_schedule_*.log)
        constexpr bool SPLIT = MT == 1 && !(YV3_PPX & 4);
        // ordered SPLIT (fp16 planes): the second k-step's fragments are requested in the order its MFMAs consume them (x_hi, w_lo..,
        // x_lo, w_hi..) and hipcc waits for them one by one (partial lgkmcnt: it can, now that the load segment's wait is the builtin):
        // a wave's ds_read_b128 return 1 KB per ~36 cycles (tools/probes/lds_read_rate.hip), so six of them take ~320 cycles -- longer
        // than the first k-step's six MFMAs
        constexpr bool OSPLIT = SPLIT && NP == 2 && !(YV3_PPX & 8);
        // ... the first OS_EARLY of them already in the load segment (then the load segment's 8 reads take about as long as the partner's 12
        // MFMAs, and the 4 left for the compute segment have returned before the second k-step starts)
        constexpr int OS_EARLY = 2;
        auto os_order = [](int q) -> int {          // fragment index (read_frag) of the q-th operand in consumption order
            if (q < MT) return NT * NP + q * NP;                       // x_hi[j]
            q -= MT;
            if (q < NT) return q * NP + 1;                             // w_lo[i]
            q -= NT;
            if (q < MT) return NT * NP + q * NP + 1;                   // x_lo[j]
            q -= MT;
            return q * NP;                                             // w_hi[i]
        };
#ifdef YV3_TIMELINE
        unsigned long long tl_load = 0, tl_b1 = 0, tl_comp = 0, tl_b2 = 0, tl_pro = 0, tl_epi = 0, tl_t = tl_entry;
        int tl_items = 0, tl_chunks = 0;
#define TL_MARK(acc_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - tl_t; tl_t = t_; } while (0)
#else
#define TL_MARK(acc_) do {} while (0)
#endif
        // ---- work list.  One tile, all of K, per workgroup -- or (SK, "stream-K") a persistent workgroup per CU that
        // owns a contiguous range [it, it_end) of (tile, K chunk) iterations of ITS XCD's tile range, so that every CU
        // gets the same number of chunks whatever the tile count (no idle CUs in a last partial round).  A range starts
        // with the TAIL part [k0, nk) of a tile (accumulators dumped to the workspace) and ends with the HEAD part
        // [0, k1) of another, whose tail the next workgroup of the same XCD dumped at ITS start, long ago: head + tail
        // are added in that fixed order and the epilogue runs once.  Ranges are >= nk chunks (host guarantees tiles >=
        // workgroups), so a tile is never split three ways.
        // The schedulable unit of the stream-K ranges is one K chunk -- or (WINO) one whole transform position (UC chunks): a
        // Winograd tile is only split between positions, where the product accumulators are zero and a part is the four
        // partial outputs.
        const int UC = WINO ? p.Cin / PBK : 1;                                    // chunks per unit
        const int UN = WINO ? 16 : p.nk;                                          // units per tile
        int it = 0, it_end = UN, tile_base = 0;
        const int jb = blockIdx.x >> 3, nj = gridDim.x >> 3;                     // (SK) this workgroup's slot in its XCD
        long long tx = 0;                                                        // (SK) unit iterations of this XCD
        if constexpr (SK) {
            const int xcd = blockIdx.x & 7;
            const int t0 = (int)((long long)xcd * p.total / 8), t1 = (int)((long long)(xcd + 1) * p.total / 8);
            tx = (long long)(t1 - t0) * UN;
            it = (int)(tx * jb / nj); it_end = (int)(tx * (jb + 1) / nj); tile_base = t0;
        }
        bool first_item = true;
        while (it < it_end) {
            int u0 = 0, u1 = UN;
            if constexpr (SK) {
                const int tl = it / UN;
                u0 = it - tl * UN;
                u1 = it_end - it < UN - u0 ? u0 + (it_end - it) : UN;
                setup_tile(tile_base + tl, u0 * UC);
                if (!first_item) __syncthreads();                                 // the previous item's epilogue tiles are dead
            }
            const int k0 = u0 * UC, k1 = u1 * UC;
            if constexpr (WINO) {
                wleft = UC; wxi = u0;
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int j = 0; j < MT; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) yac[o][i][j][e] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (k0 + d < k1) {
                    dma_prepare(k0 + d, d);
#pragma unroll
                    for (int g = 0; g < G; ++g) dma_piece(g);
                }
            // chunk k0 has landed.  After an epilogue its stores sit behind these pieces in the (load/store-mixed) vmcnt
            // queue, so later items drain it completely.
            if (first_item && k0 + D <= k1) wait_vmcnt<(D - 1) * G>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (grp == 1) __builtin_amdgcn_s_barrier();                           // group 1 runs one segment behind
            TL_MARK(tl_pro);
            int cur = 0, nxt = D % NSTAGE;
            for (int kc = k0; kc < k1; ++kc) {
                // ---- load segment: fragments of chunk kc -> registers; addresses of chunk kc+D (VALU, off the matrix
                // pipe's critical path).  My pieces of chunk kc+1 (issued D-1 compute segments ago) must have landed.
                st = lds + cur * STAGE;
#pragma unroll
                for (int ks = 0; ks < (SPLIT ? 1 : KS); ++ks)
#pragma unroll
                    for (int f = 0; f < NF; ++f) read_frag(ks, f);
                if constexpr (OSPLIT) {                                            // the second k-step's first operands: x_hi, w_lo of weight tile 0
#pragma unroll
                    for (int q = 0; q < OS_EARLY; ++q) read_frag(1, os_order(q));
                }
                const bool more = kc + D < k1;
                if (more) dma_prepare(kc + D, nxt);
                if (kc + D - 1 < k1) wait_vmcnt<(D - 2) * G>(); else wait_vmcnt<0>();
                wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
                TL_MARK(tl_load);
                __builtin_amdgcn_s_barrier();
                TL_MARK(tl_b1);
                __builtin_amdgcn_sched_barrier(0);
                // ---- compute segment: MFMAs from registers, the DMA pieces of chunk kc+D between them
                // (SPLIT: plus the second k-step's fragment reads)
                if constexpr ((YV3_PPX & 1) != 0) __builtin_amdgcn_s_setprio(1);
                if constexpr (OSPLIT) {
                    static_assert(!OSPLIT || KS == 2, "ordered SPLIT: two k-steps");
#pragma unroll
                    for (int q = OS_EARLY; q < NF; ++q) { read_frag(1, os_order(q)); __builtin_amdgcn_sched_barrier(0); }
                } else if constexpr (SPLIT) {
#pragma unroll
                    for (int ks = 1; ks < KS; ++ks)
#pragma unroll
                        for (int f = 0; f < NF; ++f) read_frag(ks, f);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (SPLIT && !OSPLIT && ks == 1) { wait_lgkmcnt<0>(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                    for (int t = 0; t < NMF; ++t)                                  // rotate over the accumulators: no back-to-back RAW
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            const int i = u / MT, j = u % MT;
                            const bf16x8v* wf = &frag[ks][i * NP];
                            const bf16x8v* xf = &frag[ks][NT * NP + j * NP];
                            if constexpr (WINO && (YV3_WABL & 8)) { asm volatile("" :: "v"(wf[t == 0 ? 1 : 0]), "v"(xf[t == 1 ? 1 : 0])); }
                            else if constexpr (NP == 2) acc[i][j] = PlaneOps<2>::mfma(wf[t == 0 ? 1 : 0], xf[t == 1 ? 1 : 0], acc[i][j]);
                            else acc[i][j] = PlaneOps<1>::mfma(wf[0], xf[0], acc[i][j]);
                            constexpr int TOT = KS * NMF * NU;                     // one DMA piece after every (TOT / G)-th MFMA
                            const int mi = (ks * NMF + t) * NU + u;
                            if (more && !(WINO && (YV3_WABL & 1)) && (mi * G) / TOT != ((mi + 1) * G) / TOT) dma_piece((mi * G) / TOT);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr ((YV3_PPX & 1) != 0) __builtin_amdgcn_s_setprio(0);
                if constexpr (WINO) {
                    if (--wleft == 0 && !(YV3_WABL & 4)) {                         // end of a transform position: Y += (A^T x A^T)[.][xi] * M
                        wleft = p.Cin / PBK;
                        const int xr = wxi >> 2, xc = wxi & 3;
                        ++wxi;
                        const float r0 = xr < 3 ? 1.f : 0.f, r1 = xr == 0 ? 0.f : (xr == 1 ? 1.f : -1.f);
                        const float q0 = xc < 3 ? 1.f : 0.f, q1 = xc == 0 ? 0.f : (xc == 1 ? 1.f : -1.f);
                        const float sc[4] = {r0 * q0, r0 * q1, r1 * q0, r1 * q1};
#pragma unroll
                        for (int i = 0; i < NT; ++i)
#pragma unroll
                            for (int j = 0; j < MT; ++j)
#pragma unroll
                                for (int e = 0; e < 16; ++e) {
                                    const float mv = acc[i][j][e];
#pragma unroll
                                    for (int o = 0; o < 4; ++o) yac[o][i][j][e] = fmaf(mv, sc[o], yac[o][i][j][e]);
                                    acc[i][j][e] = 0.f;
                                }
                    }
                }
                TL_MARK(tl_comp);
                // both groups pass the same number of barriers; group 1 skips its last one so that group 0 can start the
                // epilogue early -- except with SPLIT, where group 1 still reads fragments from LDS in its last segment
                if (SPLIT || !(grp == 1 && kc + 1 == k1)) __builtin_amdgcn_s_barrier();
                TL_MARK(tl_b2);
                __builtin_amdgcn_sched_barrier(0);
                cur = cur + 1 == NSTAGE ? 0 : cur + 1;
                nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
            }
            // group 0 gets here one segment before group 1 (which reads no LDS in its last segment; the epilogue's LDS
            // tiles are per wave)
            if (SPLIT && grp == 0) __builtin_amdgcn_s_barrier();
            if constexpr (SK) {
                constexpr int NO = WINO ? 4 : 1;                                 // accumulator sets of a part (WINO: the four outputs)
                constexpr int PART = NO * NT * MT * 16 * 64;                     // floats of one wave's part
                if (u0 > 0) {                                                    // tail / middle part: hand the accumulators over
                    float* w = p.ws + ((size_t)blockIdx.x * NW + wid) * PART + lane * 4;
#pragma unroll
                    for (int o = 0; o < NO; ++o)
#pragma unroll
                        for (int i = 0; i < NT; ++i)
#pragma unroll
                            for (int j = 0; j < MT; ++j)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const f32x16& a_ = WINO ? yac[o][i][j] : acc[i][j];
                                    *reinterpret_cast<f32x4*>(w + (((o * NT + i) * MT + j) * 4 + g) * 256) =
                                        f32x4{a_[4 * g], a_[4 * g + 1], a_[4 * g + 2], a_[4 * g + 3]};
                                }
                    // The two halves of a split tile run on the SAME XCD (workgroups b and b+8), i.e. behind the same L2:
                    // once the stores have been acknowledged (vmcnt 0: the vector L1 is write-through) the partner can
                    // read them -- no L2 write-back / invalidate (an agent-scope release/acquire pair costs ~60 us per
                    // launch here).  The XCC id travels with the flag and is checked by the reader.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(p.wsflags + blockIdx.x, 1 + (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15),
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    it += u1 - u0; first_item = false;
                    continue;
                }
                if (u1 < UN) {
                    // head part: add, in workgroup order, what the following workgroups of this XCD accumulated for the
                    // rest of this tile (one tail part, preceded by middle parts when a range is shorter than a tile)
                    const long long tile_end = (long long)(it / UN + 1) * UN;
                    for (int k = 1; jb + k < nj; ++k) {
                        const long long s_k = tx * (jb + k) / nj, e_k = tx * (jb + k + 1) / nj;
                        if (s_k >= tile_end) break;
                        if (e_k == s_k) continue;
                        const int partner = blockIdx.x + 8 * k;
                        if (tid == 0) {
                            int n = 0, f;
                            while ((f = __hip_atomic_load(p.wsflags + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++n < (1 << 22))
                                __builtin_amdgcn_s_sleep(2);
                            // never expected (reported by the host as an error): hand-over timed out, or the partner ran
                            // on another XCD, whose L2 this one is not coherent with
                            if ((f == 0 || f != 1 + (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15)) && p.flags) atomicOr(p.flags, 2);
                            __hip_atomic_store(p.wsflags + partner, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        __syncthreads();
                        const float* w = p.ws + ((size_t)partner * NW + wid) * PART + lane * 4;
#pragma unroll
                        for (int o = 0; o < NO; ++o)
#pragma unroll
                            for (int i = 0; i < NT; ++i)
#pragma unroll
                                for (int j = 0; j < MT; ++j)
#pragma unroll
                                    for (int g = 0; g < 4; ++g) {
                                        const f32x4 t = *reinterpret_cast<const f32x4*>(w + (((o * NT + i) * MT + j) * 4 + g) * 256);
                                        f32x16& a_ = WINO ? yac[o][i][j] : acc[i][j];
#pragma unroll
                                        for (int q = 0; q < 4; ++q) a_[4 * g + q] += t[q];
                                    }
                    }
                }
            }
            if constexpr (WINO) {
                if constexpr (MT == 1 && NP == 2 && YV3_WINO_EPI4) epilogue_store_wino4<BM, BN, WM, WN>(yac, p, lds, m0, n0, wid, lane);
                else {
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        epilogue_store<NP, BM, BN, WM, WN, false, false, EMTG, true>(yac[o], p, lds, m0, n0, wid, lane, o >> 1, o & 1);
                }
            } else epilogue_store<NP, BM, BN, WM, WN, OUT_F32, false, EMTG>(acc, p, lds, m0, n0, wid, lane);
            TL_MARK(tl_epi);
#ifdef YV3_TIMELINE
            ++tl_items; tl_chunks += k1 - k0;
#endif
            it += u1 - u0; first_item = false;
        }
#ifdef YV3_TIMELINE
        if (blockIdx.x == 100 && lane == 0 && p.alpha) {     // debug build only: cycle split of one workgroup -> alpha[0..]
            float* dbg = const_cast<float*>(p.alpha) + wid * 8;
            dbg[0] = (float)tl_pro / tl_items; dbg[1] = (float)tl_load / tl_chunks; dbg[2] = (float)tl_b1 / tl_chunks;
            dbg[3] = (float)tl_comp / tl_chunks; dbg[4] = (float)tl_b2 / tl_chunks; dbg[5] = (float)tl_epi / tl_items;
            dbg[6] = (float)tl_items; dbg[7] = (float)(tl_t - tl_entry);
        }
#endif
        return;
    }
    if constexpr (ROLL) {
        static_assert(KS == 2, "rolling loop: two k-steps per chunk");
        // chunk 0 has landed and is visible; its first k-step's fragments go to registers
        if (D <= p.nk) wait_vmcnt<(D - 1) * G>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int f = 0; f < NF; ++f) read_frag(0, f);
        static_assert(NP <= 2, "rolling loop: one or two planes");
        constexpr int NMF = NP == 2 ? 3 : 1;          // MFMAs per (weight tile, pixel tile, k-step)
        constexpr int TOT = NMF * NU;                 // MFMAs of one k-step block
        constexpr int RH = TOT >= 4 ? TOT / 2 : TOT;  // the other buffer's NF fragment reads follow the first RH of them
        // MFMAs of k-step `ks` (registers frag[ks]), rotating over the accumulators; `rd`: fragment reads into frag[ko] from `st`
        // behind them; `pieces`: this chunk's DMA pieces too
        auto kstep_block = [&](auto ks_c, auto ko_c, bool rd, bool pieces) {
            constexpr int ks = decltype(ks_c)::value, ko = decltype(ko_c)::value;
#pragma unroll
            for (int t = 0; t < NMF; ++t)
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int i = u / MT, j = u % MT;
                    const bf16x8v* wf = &frag[ks][i * NP];
                    const bf16x8v* xf = &frag[ks][NT * NP + j * NP];
                    if constexpr (NP == 2) acc[i][j] = PlaneOps<2>::mfma(wf[t == 0 ? 1 : 0], xf[t == 1 ? 1 : 0], acc[i][j]);
                    else acc[i][j] = PlaneOps<1>::mfma(wf[0], xf[0], acc[i][j]);
                    const int mi = t * NU + u;
                    if (rd && mi < RH) {
#pragma unroll
                        for (int f = mi * NF / RH; f < (mi + 1) * NF / RH; ++f) read_frag(ko, f);
                    }
                    if (pieces) {
#pragma unroll
                        for (int g = mi * G / TOT; g < (mi + 1) * G / TOT; ++g) dma_piece(g);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
#ifdef YV3_TIMELINE
        unsigned long long rl_prep = 0, rl_b0 = 0, rl_wait = 0, rl_bar = 0, rl_b1 = 0, rl_fold = 0, rl_t = __builtin_amdgcn_s_memtime();
        const unsigned long long rl_t0 = rl_t;       // loop entry (everything before: prologue)
#define RL_MARK(acc_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - rl_t; rl_t = t_; } while (0)
#else
#define RL_MARK(acc_) do {} while (0)
#endif
        int cur = 0, nxt = D % NSTAGE;
        for (int kc = 0; kc < p.nk; ++kc) {
            const bool more = kc + D < p.nk;
            const bool next = kc + 1 < p.nk;
            st = lds + cur * STAGE;
            if (more) dma_prepare(kc + D, nxt);
            __builtin_amdgcn_sched_barrier(0);
            RL_MARK(rl_prep);
            // ---- first k-step: MFMAs on frag[0]; behind the first half of them the second k-step's fragment reads, behind all of
            // them the DMA pieces of chunk kc+D.  (Reads go out AFTER an MFMA, never before the block's first one: the wait
            // hipcc puts in front of a block's first MFMA is lgkmcnt(0), which must not catch reads issued for the next block.)
            kstep_block(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, true, more);
            RL_MARK(rl_b0);
            // ---- the chunk's barrier: my pieces of chunk kc+1 have landed, my reads of this chunk's stage are complete
            if (next) {
                if (more) wait_vmcnt<(D - 1) * G>(); else wait_vmcnt<0>();
                wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
                RL_MARK(rl_wait);
                __builtin_amdgcn_s_barrier();
                RL_MARK(rl_bar);
                __builtin_amdgcn_sched_barrier(0);
                st = lds + (cur + 1 == NSTAGE ? 0 : cur + 1) * STAGE;
            }
            // ---- second k-step: MFMAs on frag[1]; behind the first half of them the NEXT chunk's first-k-step fragment reads
            kstep_block(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, next, false);
            RL_MARK(rl_b1);
            if constexpr (WINO) {
                if (--wleft == 0) {                                                // end of a transform position: Y += (A^T x A^T)[.][xi] * M
                    wleft = p.Cin / PBK;
                    const int xr = wxi >> 2, xc = wxi & 3;
                    ++wxi;
                    const float r0 = xr < 3 ? 1.f : 0.f, r1 = xr == 0 ? 0.f : (xr == 1 ? 1.f : -1.f);
                    const float q0 = xc < 3 ? 1.f : 0.f, q1 = xc == 0 ? 0.f : (xc == 1 ? 1.f : -1.f);
                    const float sc[4] = {r0 * q0, r0 * q1, r1 * q0, r1 * q1};
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int j = 0; j < MT; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const float mv = acc[i][j][e];
#pragma unroll
                                for (int o = 0; o < 4; ++o) yac[o][i][j][e] = fmaf(mv, sc[o], yac[o][i][j][e]);
                                acc[i][j][e] = 0.f;
                            }
                }
            }
            RL_MARK(rl_fold);
            cur = cur + 1 == NSTAGE ? 0 : cur + 1;
            nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
        }
#ifdef YV3_TIMELINE
        if (blockIdx.x == 100 && lane == 0 && p.alpha && WINO) {     // debug build only: cycle split of one workgroup -> alpha[0..]
            float* dbg = const_cast<float*>(p.alpha) + wid * 8;
            const float n_ = (float)p.nk;
            dbg[0] = rl_prep / n_; dbg[1] = rl_b0 / n_; dbg[2] = rl_wait / n_; dbg[3] = rl_bar / n_; dbg[4] = rl_b1 / n_; dbg[5] = rl_fold / n_;
            dbg[6] = n_; dbg[7] = (float)(rl_t - tl_entry);
        }
#endif
        if constexpr (WINO) {
            __syncthreads();                                                       // every wave is done with the last stage
            if constexpr (MT == 1 && NP == 2 && YV3_WINO_EPI4) epilogue_store_wino4<BM, BN, WM, WN>(yac, p, lds, m0, n0, wid, lane);
            else {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    epilogue_store<NP, BM, BN, WM, WN, false, false, EMTG, true>(yac[o], p, lds, m0, n0, wid, lane, o >> 1, o & 1);
            }
        } else epilogue_store<NP, BM, BN, WM, WN, OUT_F32, true, EMTG>(acc, p, lds, m0, n0, wid, lane);
#ifdef YV3_TIMELINE
        if (blockIdx.x == (unsigned)(p.tune[2] > 0 ? p.tune[2] : 100) && lane == 0 && p.alpha && !WINO && NW * 10 <= p.Cout) {   // non-Winograd rolling tiles (bf16): + prologue / epilogue
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            float* dbg = const_cast<float*>(p.alpha) + wid * 10;
            const float n_ = (float)p.nk;
            dbg[0] = rl_prep / n_; dbg[1] = rl_b0 / n_; dbg[2] = rl_wait / n_; dbg[3] = rl_bar / n_; dbg[4] = rl_b1 / n_; dbg[5] = n_;
            dbg[6] = (float)(rl_t0 - tl_entry); dbg[7] = (float)(t_end - rl_t); dbg[8] = (float)(t_end - tl_entry); dbg[9] = 0.f;
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
    epilogue_store<NP, BM, BN, WM, WN, OUT_F32, true, EMTG>(acc, p, lds, m0, n0, wid, lane);
}

template <int NP, int BM, int BN, int WM, int WN, int NSTAGE, int MINW = 1, int MTG = 0, bool ROLL = false>
int launch_cfg(const ConvParamsP& p, bool k3, bool dual, bool out_f32, bool use_pp, hipStream_t s) {
    const int mtiles = (p.M + BM - 1) / BM;
    const dim3 grid((unsigned)(mtiles * p.ntiles));
    const dim3 block(64 * WM * WN);
    const size_t pipe = (size_t)NSTAGE * NP * (BM + BN) * ROWB;
    const size_t epi = (size_t)WM * WN * (MTG ? MTG * 32 : BM / WM) * (BN / WN + 4) * 4;   // WM*WN waves x rows per round x (BN/WN + 4) floats
    const size_t lds = pipe > epi ? pipe : epi;
    const bool use_sk = true;                                      // stream-K persistent schedule iff the caller gave a workspace
    const int num_cu = yv3_num_cu();                               // of the CURRENT device; multiple of 8: equal workgroups per XCD
    ConvParamsP q = p;
    q.total = (int)grid.x;
    // stream-K: opt-in (the caller passes yv3_conv_desc.workspace): a split tile is summed as head + middle.. + tail, so its rounding
    // depends on where the split falls, i.e. on the batch size / the image's position in the batch -- results stay
    // within the parity tolerance but are no longer bit-identical across batch compositions.
    // It is used only for launches of fewer than two rounds of tiles (the 13x13 layers at bs=64; nearly every layer of a
    // small batch, where it splits each tile's K range over the otherwise idle CUs): filling the idle CUs of a last
    // partial round buys nothing on this power-limited kernel (the busy CUs simply clock higher: measured -9 % on the
    // 2.6- and 5.3-round layers, which also lose the hardware's dynamic tile dispatch), but with 1.3 rounds the even
    // split wins, and it lets the 13x13 3x3 layers use 256x128 tiles (+9 ... +11 %).
    // Measured rule (tools/conv_bench.py, bs = 4 ... 64): it pays for the long-K 3x3 layers when the tiles fill 1 - 2
    // rounds (even split instead of a 30 - 100 % idle second round) or at most 0.4 rounds (each tile's K range spread
    // over the idle CUs: the 13x13 3x3 layer at bs=4 0.084 -> 0.039 ms); it loses for 1x1 layers (the accumulator
    // exchange outweighs their few K chunks) and around 0.7 rounds.
    const bool sk_shape = k3 && ((q.total >= num_cu && q.total < 2 * num_cu) || 5 * q.total <= 2 * num_cu);
    const bool sk = use_sk && p.ws && p.wsflags && sk_shape && (long long)q.total * p.nk >= num_cu &&
                    num_cu <= YV3_SK_MAX_WG && p.ws_bytes >= yv3_conv_workspace_bytes();
    const dim3 sgrid((unsigned)num_cu);
#define YV3_LAUNCH(K3_, DUAL_, OF_) do { \
    if constexpr (NP == 1 && WM * WN == 8 && NSTAGE >= 3 && !ROLL) { \
        if (use_pp) { hipLaunchKernelGGL((conv_planes_kernel<NP, BM, BN, WM, WN, NSTAGE, K3_, DUAL_, OF_, true, false, 1, MTG>), grid, block, lds, s, q); break; } \
    } \
    if constexpr (NP == 2 && WM * WN == 8 && NSTAGE >= 3 && !ROLL) { \
        if (use_pp && sk) { hipLaunchKernelGGL((conv_planes_kernel<NP, BM, BN, WM, WN, NSTAGE, K3_, DUAL_, OF_, true, true>), sgrid, block, lds, s, q); break; } \
        if (use_pp) { hipLaunchKernelGGL((conv_planes_kernel<NP, BM, BN, WM, WN, NSTAGE, K3_, DUAL_, OF_, true>), grid, block, lds, s, q); break; } \
    } \
    hipLaunchKernelGGL((conv_planes_kernel<NP, BM, BN, WM, WN, NSTAGE, K3_, DUAL_, OF_, false, false, MINW, MTG, false, ROLL>), grid, block, lds, s, q); } while (0)
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
int yv3_conv2d_planes_w4(const ConvParamsP* pp, int np, int npad, hipStream_t s);
int yv3_wino_input_transform(const u16* x, long long xs, u16* v, int B, int H, int W, int C, hipStream_t s);

// Winograd F(2x2,3x3) form of a 3x3 / stride-1 fp16-plane layer: input transform (winograd.hip) + the 16-position GEMM with the
// output transform folded into the main loop.  `p` = the direct launch's parameters (x, res, y, strides, Cout, act, flags).
static int launch_wino(const yv3_conv_desc* d, ConvParamsP p, hipStream_t s) {
    const int th = (d->H + 1) / 2, tw = (d->W + 1) / 2;
    const long long T = (long long)d->B * th * tw;
    if (T > 0x7fffffffLL || d->cout_pad % 128 || d->cin % 32) return YV3_ESHAPE;
    if (!d->wino_ws || d->wino_ws_bytes < (size_t)2 * 16 * T * d->cin * sizeof(u16)) return YV3_EWORKSPACE;
    u16* v = (u16*)d->wino_ws;
    int rc = yv3_wino_input_transform(p.x, p.xs, v, d->B, d->H, d->W, d->cin, s);
    if (rc) return rc;
    p.x = v; p.xs = 16 * T * d->cin; p.xi_stride = T * d->cin;
    p.w = (const u16*)d->w_wino; p.alpha = d->alpha_wino;
    p.wH = d->H; p.wW = d->W; p.wth = th; p.wtw = tw;
    p.H = 1; p.W = (int)T; p.Ho = 1; p.Wo = (int)T; p.M = (int)T; p.stride = 1;
    p.K = 16 * d->cin; p.nk = p.K / PBK;
    p.tb = 128; p.ntiles = d->cout_pad / 128;
    constexpr int BM = 128, BN = 128, NS = 4;
    const dim3 grid((unsigned)(((T + BM - 1) / BM) * p.ntiles));
    p.total = (int)grid.x;
    const size_t pipe = (size_t)NS * 2 * (BM + BN) * ROWB, epi = (size_t)8 * 32 * (BN / 2 + 4) * 4;
    const size_t lds = pipe > epi ? pipe : epi;
    // Schedules.  Default: one 128x128 tile (all 16 positions, 16 x Cin/32 chunks) per workgroup -- bitwise independent of the
    // batch composition.  YV3_OPT_WINO_EVEN: stream-K over transform positions -- one persistent workgroup per CU takes an equal,
    // contiguous range of (tile, position) units of its XCD and hands partial outputs over inside the XCD's L2 (see the
    // kernel); a split tile is summed head + tail.  Measured (tools/wino_ab.py, profiles/r03_wino_ab2.log): it only wins below
    // half a round of tiles (512->1024 @19x19 bs=16: 0.176 vs 0.202 ms) and loses 3...30 % above (256 KB of partial outputs per
    // split, no dynamic tile dispatch, and a partly filled round simply clocks higher on this power-limited chip): opt-in.
    const int num_cu = yv3_num_cu();
    const size_t vbytes = (size_t)2 * 16 * T * d->cin * sizeof(u16);
    const bool even = (d->options & YV3_OPT_WINO_EVEN) && num_cu <= YV3_WINO_SK_MAX_WG && (long long)p.total * 16 >= num_cu &&
                      p.total % num_cu != 0 && d->wino_ws_bytes >= vbytes + yv3_wino_sk_bytes();
    if (even) {
        p.ws = (float*)((char*)d->wino_ws + ((vbytes + 255) & ~(size_t)255));
        p.wsflags = (int*)((char*)p.ws + (size_t)YV3_WINO_SK_MAX_WG * YV3_WINO_SK_PART_BYTES);
        p.ws_bytes = yv3_wino_sk_bytes();
        hipLaunchKernelGGL((conv_planes_kernel<2, BM, BN, 4, 2, NS, false, false, false, true, true, 1, 0, true>), dim3((unsigned)num_cu), dim3(512), lds, s, p);
    } else {
        p.ws = nullptr; p.wsflags = nullptr; p.ws_bytes = 0;
        // two-group ping-pong loop (default) or the rolling single-phase loop (one barrier per chunk, fragment reads spread under the
        // MFMAs; tune[1] bit 1: A/B measurements -- bit-identical, equal speed: profiles/r04d_wino_roll_vs_pingpong_ab.log; the same stage
        // on FOUR waves with 64x64 wave tiles and the rolling loop was 1.4x slower, profiles/r04f_wino_4waves_roll_ab.log)
#ifndef YV3_WINO_ROLL
#define YV3_WINO_ROLL 0
#endif
        if (((p.tune[1] >> 1) & 1) != YV3_WINO_ROLL)
            hipLaunchKernelGGL((conv_planes_kernel<2, BM, BN, 4, 2, NS, false, false, false, false, false, 1, 0, true, true>), grid, dim3(512), lds, s, p);
        else
            hipLaunchKernelGGL((conv_planes_kernel<2, BM, BN, 4, 2, NS, false, false, false, true, false, 1, 0, true>), grid, dim3(512), lds, s, p);
    }
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t yv3_conv_workspace_bytes(void) { return (size_t)YV3_SK_MAX_WG * (YV3_SK_PART_BYTES + sizeof(int)); }

static int planes_wino_rule(const yv3_conv_desc* d, int np);

// The shape errors yv3_conv2d_planes reports before it launches anything: ONE function, called by the launch path and by the
// form query, so that yv3_conv2d_form returns exactly "the YV3_E* code yv3_conv2d would return" (include/yv3.h).
static int planes_shape_rc(const yv3_conv_desc* d) {
    if (d->dec_out && (d->out_dtype != YV3_F32 || d->cout % 3 || d->dec_stride <= 0.f)) return YV3_ESHAPE;
    const int pad = (d->k - 1) / 2;
    const long long Ho = (d->H + 2 * pad - d->k) / d->stride + 1, Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
    if ((long long)d->B * Ho * Wo > 0x7fffffffLL) return YV3_ESHAPE;
    if (d->out_dtype != YV3_F32 && (d->cout % 8)) return YV3_ESHAPE;
    const int npad = d->cout_pad, tb = npad < 128 ? npad : 128;
    if (tb <= 0 || npad % tb) return YV3_ESHAPE;
    if (d->out_dtype == YV3_F32 && (d->k == 3 || d->cin_up > 0)) return YV3_ESHAPE;      // fp32 outputs are the 1x1 head convs
    return 0;
}

// Does the opt-in kw-tap-reuse kernel (YV3_OPT_K3S1, conv_planes_k3s1.hip) take this launch?  It is dispatched BEFORE the Winograd rule.
static bool k3s1_takes(const yv3_conv_desc* d) {
    return d->k == 3 && d->stride == 1 && d->out_dtype != YV3_F32 && (d->options & YV3_OPT_K3S1) && d->cin % PBK == 0 &&
           (d->cout_pad % 128 == 0 || d->cout_pad % 64 == 0);
}

// Which form does this descriptor take?  (The per-launch rule of the fp16-plane mode; also exported through yv3_conv2d_form so
// that callers -- tests, bench.py's executed-FLOP accounting -- see the choice the library makes.)  Mirrors yv3_conv2d_planes'
// dispatch order: shape errors, the opt-in k3s1 kernel (direct form), the Winograd rule (+ launch_wino's own errors).
int yv3_conv2d_planes_form(const yv3_conv_desc* d, int np) {
    const int src = planes_shape_rc(d);
    if (src) return src;
    if (k3s1_takes(d)) return YV3_FORM_DIRECT;
    const int w = planes_wino_rule(d, np);
    if (w == 1) {
        const long long T = (long long)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2);
        if (T > 0x7fffffffLL || d->cout_pad % 128 || d->cin % 32) return YV3_ESHAPE;
        if (!d->wino_ws || d->wino_ws_bytes < (size_t)2 * 16 * T * d->cin * sizeof(u16)) return YV3_EWORKSPACE;
    }
    return w;
}

static int planes_wino_rule(const yv3_conv_desc* d, int np) {
    const int npad = d->cout_pad;
    const bool k3 = d->k == 3, dual = d->cin_up > 0, out_f32 = d->out_dtype == YV3_F32;
    if (!(d->w_wino && np == 2 && k3 && d->stride == 1 && !out_f32 && !dual && d->alpha_wino && npad % 128 == 0 &&
          d->x_plane_stride <= 0 && d->y_plane_stride <= 0)) return 0;
    // Winograd F(2x2,3x3) when its 128x128 tiles (a quarter of the direct kernel's row count) fill 0.55 ... 1.05 rounds of
    // the chip: same-box A/B against the direct kernel (tools/wino_ab.py): 256->512 @26x26 bs=32 (172 tiles) x1.28,
    // 512->1024 @13x13 bs=64 (200) x1.36, 256->512 @38x38 bs=16 (184) x1.26; but 340 tiles (1.33 rounds: @26x26 bs=64) x0.96,
    // 104 tiles (@13x13 bs=32, @19x19 bs=16) x0.78...0.80, and the 128-channel 52x52 layers x0.93 (input transform HBM-bound)
    const long long tiles = (((long long)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2) + 127) / 128) * (npad / 128);
    const long long ncu = yv3_num_cu();
    // Round 3, later (tools/wino_ab.py over bs = 48 ... 256, profiles/r03x_wino_rounds_map.log): what decides is how full the LAST
    // round of tiles is.  r = tiles / CUs: 0.59 x0.89, 0.78 x1.38, 0.97 x1.27, 1.00 x1.15 | 1.16 x0.85, 1.33 x0.94, 1.53 x1.06, 1.66 x1.16,
    // 1.94 x1.29, 2.31 x1.01, 2.64 x1.14, 3.06 x1.07, 3.97 x1.13, 5.28 x1.09.  Rule for a launch that has the chip to itself:
    // up to one round r >= 0.62; beyond, r / ceil(r) >= 0.75.  Under two concurrent lanes (YV3_OPT_TWO_LANES) the other lane's
    // launch fills the idle part of a round: r >= 0.27 (the 13x13 layers at 32 images per lane, 104 tiles, run x0.78 alone but the
    // two-lane step gains 2.6-3.8 % with them; at 64 / 128 images per lane the 1.33-round 26x26 layers gain too: bs=128 +2.9 %, bs=256
    // +5.5 %, profiles/r03y_wino_two_lanes_rule_ab.txt, r03x_wino_big_batch.txt)
    if (d->options & YV3_OPT_WINO_ALWAYS) return 1;
#ifndef YV3_AB_NO_TWO_LANES_RULE
    if (d->options & YV3_OPT_TWO_LANES) return tiles * 100 >= 27 * ncu;
#endif
    if (tiles * 100 <= 105 * ncu) return tiles * 100 >= 62 * ncu;
    return tiles * 100 >= 75 * ((tiles + ncu - 1) / ncu) * ncu;
}

int yv3_conv2d_planes(const yv3_conv_desc* d, int np, hipStream_t s) {
    const int src = planes_shape_rc(d);
    if (src) return src;
    ConvParamsP p;
    p.x = (const u16*)d->x; p.x2 = (const u16*)d->x2; p.w = (const u16*)d->w;
    p.alpha = d->alpha; p.beta = d->beta; p.res = (const u16*)d->residual; p.y = d->y;
    p.H = d->H; p.W = d->W; p.Cin = d->cin; p.Cup = d->cin_up; p.Cout = d->cout;
    p.stride = d->stride; p.act = d->act; p.flags = d->flags;
    for (int i = 0; i < 4; ++i) p.tune[i] = d->tune[i];
    p.dec_out = nullptr;
    if (d->dec_out) {
        if (d->out_dtype != YV3_F32 || d->cout % 3 || d->dec_stride <= 0.f) return YV3_ESHAPE;
        p.dec_out = d->dec_out; p.dec_bs = d->dec_out_batch_stride; p.dec_stride = d->dec_stride;
        for (int i = 0; i < 6; ++i) p.dec_an[i] = d->dec_anchors[i] / d->dec_stride;     // float32 division, as torch does
    }
    p.ws = (float*)d->workspace; p.ws_bytes = d->workspace ? d->workspace_bytes : 0;
    p.wsflags = d->workspace ? (int*)((char*)d->workspace + (size_t)YV3_SK_MAX_WG * YV3_SK_PART_BYTES) : nullptr;
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
    if (d->x_plane_stride > 0) p.xs = d->x_plane_stride;            // batch slices of larger plane tensors
    if (d->x2_plane_stride > 0) p.x2s = d->x2_plane_stride;
    if (d->y_plane_stride > 0) p.ys = d->y_plane_stride;
    const bool out_f32 = d->out_dtype == YV3_F32;
    if (!out_f32 && (d->cout % 8)) return YV3_ESHAPE;
    const int npad = d->cout_pad;
    p.tb = npad < 128 ? npad : 128;
    if (npad % p.tb) return YV3_ESHAPE;
    const bool k3 = d->k == 3, dual = d->cin_up > 0;
    // kw-tap reuse kernel (conv_planes_k3s1.hip): 44 % less L2->LDS traffic, same results, but no faster on
    // MI355X because this MFMA stream is power-limited -- opt-in until that changes.
    if (k3s1_takes(d)) {
        const int rc = yv3_conv2d_planes_k3s1(&p, np, npad, M, s);
        if (rc != -100) return rc;
    }
    if (planes_wino_rule(d, np) == 1) return launch_wino(d, p, s);
    const bool use_pp = !(d->options & YV3_OPT_NO_PINGPONG);       // ping-pong main loop (fp16x2, 8-wave tiles) unless disabled
#define YV3_CFG(BM_, BN_, WM_, WN_, NS_) (np == 3 ? launch_cfg<3, BM_, BN_, WM_, WN_, NS_>(p, k3, dual, out_f32, use_pp, s) : \
                                         np == 2 ? launch_cfg<2, BM_, BN_, WM_, WN_, (NS_) + 1>(p, k3, dual, out_f32, use_pp, s) : \
                                                   launch_cfg<1, BM_, BN_, WM_, WN_, NS_>(p, k3, dual, out_f32, false, s))
    if (npad % 128 == 0) {
        // 256x128 tiles (8 waves, 144 KB LDS; 64x64 per wave) from half a round of tiles upwards, else 128x128 tiles
        // (8 waves of 32x64).  Measured at bs=64: the 13x13 layers have 172 / 344 big tiles (0.7 / 1.3 rounds) and are
        // still 5 % (3x3) to 26 % (1x1) faster than with 340 / 680 small ones -- the big tile does 1/3 less LDS
        // traffic per MFMA, and a partly filled round simply clocks higher on this power-limited kernel.
        const long long blocks256 = ((M + 255) / 256) * (npad / 128);
        p.ntiles = npad / 128;
        // (with the stream-K schedule every CU gets the same share whatever the tile count: one tile per CU suffices)
        const bool sk_ok = np == 2 && p.ws && use_pp;
        const int big_min = d->big_tile_min > 0 ? d->big_tile_min : 128;
        const int force = (int)((d->options >> YV3_OPT_TILE_SHIFT) & 0xffu);
        // (code 12: the four-wave 256x128 tile with 16-deep chunks, two workgroups per CU -- conv_planes_w4.hip)
        if (np == 2 && force == 12 && !out_f32 && !dual) {
            const int rc = yv3_conv2d_planes_w4(&p, np, npad, s);
            if (rc != -100) return rc;
        }
        if (np == 2 && force == 3) return launch_cfg<2, 128, 128, 2, 2, 2, 2>(p, k3, dual, out_f32, false, s);
        if (np == 1 && force == 3) return launch_cfg<1, 128, 128, 2, 2, 2, 2>(p, k3, dual, out_f32, false, s);
        // 256x128 tile on FOUR waves (128x64 wave tiles: 6 fragment reads per 8 MFMAs instead of 4 per 4), two workgroups per CU
        if (np == 1 && force == 5) return launch_cfg<1, 256, 128, 2, 2, 3, 2, 2>(p, k3, dual, out_f32, false, s);
        // 256x256 tile on eight waves (128x64 wave tiles), one workgroup per CU, single-phase loop
        if (np == 1 && force == 6 && npad % 256 == 0 && !out_f32) { p.ntiles = npad / 256; return launch_cfg<1, 256, 256, 2, 4, 3, 1, 1>(p, k3, dual, out_f32, false, s); }
        // (code 8: the 256x256 tile with the rolling loop; code 9: 4-deep ring)
        if (np == 1 && force == 8 && npad % 256 == 0 && !out_f32) { p.ntiles = npad / 256; return launch_cfg<1, 256, 256, 2, 4, 3, 1, 1, true>(p, k3, dual, out_f32, false, s); }
        if (np == 1 && force == 9 && npad % 256 == 0 && !out_f32) { p.ntiles = npad / 256; return launch_cfg<1, 256, 256, 2, 4, 4, 1, 1, true>(p, k3, dual, out_f32, false, s); }
        // (code 11: the 192-row variant of the 256x256 rolling tile -- 96x64 wave tiles; also measured and dropped: 192x128 on four waves
        // and 128x256 on eight, profiles/r04aa_bf16_192row_tiles_ab.log)
        // (codes 13 / 14: the 256x256 tile with the eight-wave PING-PONG loop, 128x64 wave tiles, 3- / 4-deep ring; 14 is what the rule below ships)
        if (np == 1 && force == 13 && npad % 256 == 0 && !out_f32) { p.ntiles = npad / 256; return launch_cfg<1, 256, 256, 2, 4, 3, 1, 1>(p, k3, dual, out_f32, true, s); }
        if (np == 1 && force == 14 && npad % 256 == 0 && !out_f32) { p.ntiles = npad / 256; return launch_cfg<1, 256, 256, 2, 4, 4, 1, 1>(p, k3, dual, out_f32, true, s); }
        // (code 15: the 192-row variant with the ping-pong loop)
        if (np == 1 && force == 15 && npad % 256 == 0 && !out_f32) { p.ntiles = npad / 256; return launch_cfg<1, 192, 256, 2, 4, 3, 1, 1>(p, k3, dual, out_f32, true, s); }
        // (code 16: ... with a 4-deep ring: one more chunk of prefetch lead)
        if (np == 1 && force == 16 && npad % 256 == 0 && !out_f32) { p.ntiles = npad / 256; return launch_cfg<1, 192, 256, 2, 4, 4, 1, 1>(p, k3, dual, out_f32, true, s); }
        if (np == 1 && force == 11 && npad % 256 == 0 && !out_f32) { p.ntiles = npad / 256; return launch_cfg<1, 192, 256, 2, 4, 3, 1, 1, true>(p, k3, dual, out_f32, false, s); }
        if (np == 2 && force == 4) { p.ntiles = npad / 64; return launch_cfg<2, 128, 64, 2, 2, 2>(p, k3, dual, out_f32, false, s); }
        // Round 5: the four-wave 192x128 tile, TWO workgroups per CU (conv_planes_w4.hip): one workgroup's prologue / epilogue / launch gap
        // under the other's main loop; bit-identical to the eight-wave tile (same K order).  Same-box A/B, bs=64 (profiles/r05e_w4_192x128_ab.txt):
        // 128->256 @52 +4 %, 64->128 @104 +5 %, 512->256 1x1 @26 +12 %, 256->128 1x1 @52 +6 %, 512->1024 s2 @13 +11 %; at bs=32 256->512 @26
        // +16 %, 512->1024 @13 +9 % (192-row tiles fill the chip's last round better); 256->512 @26 bs=64 -3 %, long K (512->256 3x3 @52) -5 %.
        // (tune[1] bit 5: off, bit 6: off for 1x1 layers, bit 7: off for 3x3 layers -- A/B measurements)
        // End to end (profiles/r05h_w4_end_to_end_ab_other_batches.txt, r05j_*): 416x416 bs=16 +7.7 %, bs=32 +3.8 %, bs=8 +2.3 %, 608x608 bs=16 +4 %,
        // dense 608x608 bs=8 +3.4 %, bs=64 on one lane +0.8 % (the chip is power-limited there: 16 % more tile rows per CU-cycle by the kernel's own
        // timeline, profiles/r05f_w4_timeline.txt, buy 4 % in isolation and ~1 % in the network).  Under TWO concurrent lanes the 3x3 layers lose
        // with it (bs=64: -1.2 %; three alternating passes) while the 1x1 layers still gain (+0.2 %): there only the 1x1 layers take it.
        const bool w4_lanes_ok = !(d->options & YV3_OPT_TWO_LANES) || !k3 || (p.tune[1] & 256);
        if (np == 2 && force == 0 && !out_f32 && !dual && !(p.tune[1] & 32) && (k3 || p.nk >= 8) && !(p.tune[1] & (k3 ? 128 : 64)) && w4_lanes_ok) {
            const long long t192 = ((M + 191) / 192) * (npad / 128);
            // from three quarters of a workgroup per CU upwards (512->1024 @19x19 bs=8: 128 tiles on 256 CUs, one four-wave workgroup on
            // every other CU, 219 instead of 292 TFLOP/s; 232 tiles @13x13 bs=32: +5 %; profiles/r05i_w4_layers_*.txt)
            if (t192 * 4 >= 3 * yv3_num_cu()) {
                const int rc = yv3_conv2d_planes_w4(&p, np, npad, s);
                if (rc != -100) return rc;
            }
        }
        // short-K 1x1 layers (K <= 512: 8-16 chunks per tile, mostly prologue / epilogue): two independent 4-wave workgroups
        // per CU (128x128 tiles, 2-deep ring) hide each other's IO -- in the network at bs=64 the step gains 0.8 %
        // (13.31 -> 13.20 ms, same box, alternating; K = 1024 does not gain); same K order, same bits
        // (the head convs at 52x52 / 26x26 included: +0.2...0.7 %; the 104x104 3x3 layers, K = 576, lose 1 % on it)
        if (np == 2 && !k3 && p.nk <= 16 && force == 0 && blocks256 >= big_min && !(p.tune[1] & 1))
            return launch_cfg<2, 128, 128, 2, 2, 2, 2>(p, k3, dual, out_f32, false, s);
        if (force == 1) return YV3_CFG(256, 128, 4, 2, 2);
        if (force == 2) return YV3_CFG(128, 128, 4, 2, 3);
        // one bf16 plane (YV3_BF16): the same ping-pong loop with one MFMA per unit -- 608x608 bs=16: 2727 -> 3155
        // images/s on one lane (two 4-wave workgroups per CU instead: 2953)
        // (6-deep ring, 147 KB: with 8 MFMAs per chunk and wave a DMA piece needs several chunk times to land; 3-deep 3690, 4-deep
        // 3830, 6-deep 3870 images/s at 608x608 bs=16)
        // ... and from one tile per CU upwards the same 256x128 tile on FOUR waves (128x64 wave tiles: 6 fragment reads per 8 MFMAs
        // instead of 4 per 4, half the DMA pieces per MFMA and wave), two independent workgroups per CU (72 KB of LDS each, <= 256
        // registers), single-phase loop, epilogue in two rounds of 64 rows: same K order, bit-identical.  Same-box A/B
        // (tools/tile_ab.py, profiles/r03_bf16_tile_ab*.log), 608x608 bs=16: 128->256 @76 665 -> 772 TFLOP/s, 256->512 @38 670 -> 811,
        // 64->128 @152 559 -> 691, the stride-2 layers +11...18 %, 256->128 1x1 @76 +12 %; 416x416 bs=64: @52 663 -> 831, @26 825 -> 901,
        // @13 702 -> 846.  Below one tile per CU (512->1024 @19 at bs=16: 184 tiles, 1x1 layers at 38 / 19) the 8-wave ping-pong
        // tile wins by 7...30 % (twice the waves per tile).  A 256x256 / 8-wave tile (code 6) loses to both at these sizes.
        // (tune[2] > 0: threshold override for A/B measurements)
        // (code 7: the same tile with the rolling loop -- barrier between the k-steps, next chunk's first fragments read under the MFMAs)
        if (np == 1 && force == 7) return launch_cfg<1, 256, 128, 2, 2, 3, 2, 2, true>(p, k3, dual, out_f32, false, s);
        // Round 4 (tools/tile_ab.py, profiles/r04d_bf16_roll_ab.log, r04j_bf16_tiles_ab.log): the 3x3 layers take the ROLLING loop on that
        // tile (+2...6 %; the 1x1 layers lose 1-3 % on it and keep the plain loop) -- and a 256x256 tile on eight waves (128x64 wave tiles,
        // one workgroup per CU, rolling loop: 32 KB of DMA per 128 MFMAs instead of 24 KB per 64 -- the L2 -> LDS path delivers 62 B/clk/CU,
        // tools/probes/dma_rate.hip, and was the 256x128 tile's co-bottleneck) when its tile count fills the chip's last round:
        // 676 tiles (128->256 @52x52 bs=64) +8 %, 172 (512->1024 @13x13 bs=64) +15 %, 182 (256->512 @38x38 bs=16) +14 %; but 338
        // (1.32 rounds) -7 %, 361 -6 %, 92 -28 %.  Rule: r = tiles / CUs; r >= 0.6 up to one round, r / ceil(r) >= 0.8 beyond.
        if (np == 1 && force == 0 && k3 && !out_f32 && npad % 256 == 0 && !(p.tune[1] & 8)) {
            const long long t256 = ((M + 255) / 256) * (npad / 256);
            const long long ncu = yv3_num_cu();
            const bool fill = t256 <= ncu ? t256 * 10 >= 6 * ncu : t256 * 10 >= 8 * ((t256 + ncu - 1) / ncu) * ncu;
            // ... and its 192-row variant (96x64 wave tiles; a wave stages 24 pixel rows = one DMA piece and a half) where that fills the
            // last round to >= 85 % and 256 rows leave it below 80 %: 256->512 @38x38 bs=16 (182 -> 242 tiles) +6 %, 512->1024 @13x13 bs=64
            // (172 -> 228) +5 %, 128->256 @76x76 bs=16 (361 -> 482) +2.6 %, @76x76 bs=8 +9 % (profiles/r04aa_bf16_192row_tiles_ab.log)
            const long long t192 = ((M + 191) / 192) * (npad / 256);
            const long long r256 = (t256 + ncu - 1) / ncu * ncu, r192 = (t192 + ncu - 1) / ncu * ncu;
            // Round 5: both tiles run the eight-wave PING-PONG loop on a 4-deep ring instead of the rolling loop (tune[1] bit 9: the rolling loop,
            // bit 10: ping-pong on the 3-deep ring -- A/B).  The rolling tile's eight waves leave their one barrier together, their fragment
            // reads (96 KB per chunk and CU) queue behind each other and ~350 of a chunk's 1500 cycles are exposed LDS latency
            // (tools/timeline.py --kernel roll_bf16, profiles/r05x_bf16_roll_timeline.txt); with one four-wave group reading while the other issues
            // MFMAs the layers run bit-identical and +5...+14 % faster in isolation, on uniform random operands and on the network's own
            // activations alike (profiles/r05y_bf16_pingpong_*_ab.txt, r05ad_*).  IN the network the 3-deep ring LOSES 2 % (its DMA lead is one
            // compute segment, ~1000 cycles: fine for L2-hot repeats of one layer, too short for a layer's first touch of its weights and
            // inputs); the 4-deep ring gains: conv kernel time 608x608 bs=16 3.17 -> 3.08 ms, 416x416 bs=64 5.21 -> 5.04 ms, step +2 %
            // (profiles/r05ae_*, r05af_*; a per-layer A/B decides nothing by itself).
            const bool roll = (p.tune[1] & 512) != 0, pp3 = (p.tune[1] & 1024) != 0;
            if (!(p.tune[1] & 16) && t192 * 100 >= 85 * r192 && t256 * 100 < 80 * r256) {
                p.ntiles = npad / 256;
                if (roll) return launch_cfg<1, 192, 256, 2, 4, 3, 1, 1, true>(p, k3, dual, out_f32, false, s);
                return pp3 ? launch_cfg<1, 192, 256, 2, 4, 3, 1, 1>(p, k3, dual, out_f32, true, s) : launch_cfg<1, 192, 256, 2, 4, 4, 1, 1>(p, k3, dual, out_f32, true, s);
            }
            if (fill) {
                p.ntiles = npad / 256;
                if (roll) return launch_cfg<1, 256, 256, 2, 4, 3, 1, 1, true>(p, k3, dual, out_f32, false, s);
                return pp3 ? launch_cfg<1, 256, 256, 2, 4, 3, 1, 1>(p, k3, dual, out_f32, true, s) : launch_cfg<1, 256, 256, 2, 4, 4, 1, 1>(p, k3, dual, out_f32, true, s);
            }
        }
        if (np == 1 && force == 0 && k3 && blocks256 >= (p.tune[2] > 0 ? p.tune[2] : 256) && !out_f32 && !(p.tune[1] & 8))
            return launch_cfg<1, 256, 128, 2, 2, 3, 2, 2, true>(p, k3, dual, out_f32, false, s);
        if (np == 1 && force == 0 && blocks256 >= (p.tune[2] > 0 ? p.tune[2] : 256) && !out_f32) return launch_cfg<1, 256, 128, 2, 2, 3, 2, 2>(p, k3, dual, out_f32, false, s);
        if (np == 1 && use_pp && force == 0 && blocks256 >= big_min) return launch_cfg<1, 256, 128, 4, 2, 6>(p, k3, dual, out_f32, true, s);
        if (blocks256 >= (sk_ok ? 256 : big_min)) return YV3_CFG(256, 128, 4, 2, 2);
        return YV3_CFG(128, 128, 4, 2, 3);
    }
    if (npad % 64 == 0) {
        p.ntiles = npad / 64;
        // fp16 planes: 2-deep ring (49 KB) -> three workgroups per CU instead of two (+4 % on the 208x208 3x3 layer)
        if (np == 2) return launch_cfg<2, 128, 64, 2, 2, 2>(p, k3, dual, out_f32, false, s);
        return YV3_CFG(128, 64, 2, 2, 2);
    }
    p.ntiles = npad / 32;
    if (np == 2) return launch_cfg<2, 128, 32, 4, 1, 2>(p, k3, dual, out_f32, false, s);     // (+6 % on the 208x208 1x1 layer)
    return YV3_CFG(128, 32, 4, 1, 2);
#undef YV3_CFG
}
