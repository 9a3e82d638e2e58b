// fp16-plane convolution, 192x128 tile on FOUR waves, TWO workgroups per CU ("w4").
//
// Why another main loop (round 5).  conv_planes_kernel's 256x128 tile runs on eight waves with a 144 KB LDS ring, ONE
// workgroup per CU: a tile's prologue (index math + the first HBM round trip, 7-10 k cycles), its epilogue (12-19 k: residual
// rows in, two output planes out) and the gap until the next workgroup starts (~6 k) overlap nothing -- 26-43 % of a tile's time
// at 36 / 18 K chunks (profiles/r04an_persistent_stream_ab.txt); making the workgroup persistent hid the prologue but put every CU's
// HBM-heavy epilogue in phase (-0.5 %).  This kernel keeps the hardware's dynamic one-tile dispatch and instead makes TWO
// workgroups fit on a CU (2 x 80 KB of LDS, <= 256 registers per wave), so that one's prologue / epilogue / launch gap sits under
// the other's main loop:
//   * 192x128 tile, 32-deep K chunks, TWO ring stages of 2 planes x (192 + 128) rows x 64 B = 40 KB;
//   * four waves with 96x64 wave tiles (6 accumulator blocks): per 16-deep k-step a wave issues 10 ds_read_b128 for 18 MFMAs (the
//     eight-wave tile: 8 for 12) -- a wave's LDS reads cost it ~36 cycles of issue each (tools/probes/lds_read_rate.hip), the
//     co-limiter of the eight-wave loop;
//   * single-phase rolling loop over two fragment register sets: while the MFMAs of k-step 0 run, k-step 1's fragments are read;
//     then the chunk's ONE barrier (every wave has now taken everything it needs from this chunk's stage, and the next chunk has
//     landed in the other stage); under k-step 1's MFMAs the stage just released is refilled by DMA (chunk k+2) and the next
//     chunk's k-step-0 fragments are read.  Nothing is read or waited for right behind the barrier; what latency is still exposed
//     is filled by the other workgroup's waves.
// (First attempt, measured and dropped -- profiles/r05d_w4_16deep_ab.txt, source kept as tools/probes/dead_experiments/conv_planes_w4_16deep_chunks.hip.txt:
// 16-deep chunks with a 256x128 tile, three 24 KB stages.  Bit-identical, but global_load_lds moves 32-byte row segments at HALF the
// rate of 64-byte ones, 31 instead of 62 B/clk/CU (tools/probes/dma_rate.hip, rows32; profiles/r05d_dma_rate_rows32.txt): the DMA
// stream then takes as long as the MFMAs, long-K layer 297 instead of 377 TFLOP/s.)
// Same K order and the same three-product order per k-step as conv_planes_kernel: results are BIT-IDENTICAL to it
// (tools/tile_ab.py asserts that); the packed weights are shared.
//
// Replaces the same reference call sites as conv_planes.hip (darknet.py:43-44, :52-53): conv_bn_relu / res_layer.conv2 with
// 3x3 (stride 1 or 2) or plain 1x1 filters, plane outputs, cout_pad % 128 == 0.
#include <type_traits>
#include "conv_planes_common.h"

namespace {

constexpr int W4_BM = 192, W4_BN = 128;
constexpr int W4_NS = 2;                        // ring stages
constexpr int W4_A_PLANE = W4_BM * ROWB;        // 12 KB
constexpr int W4_B_PLANE = W4_BN * ROWB;        // 8 KB
constexpr int W4_STAGE = 2 * (W4_A_PLANE + W4_B_PLANE);      // 40 KB
constexpr int W4_AQ = W4_BM / 4 / RPG;          // pixel-side DMA wave instructions per plane and wave (48 rows = 3 x 16)
constexpr int W4_BQ = W4_BN / 4 / RPG;          // weight side (32 rows = 2 x 16)
constexpr int W4_G = 2 * (W4_AQ + W4_BQ);       // DMA wave instructions per chunk and wave (10)

#ifndef YV3_W4_SPLIT_DMA
#define YV3_W4_SPLIT_DMA 1                      // (0: all ten DMA pieces behind the first MFMAs of k-step 1 -- A/B builds)
#endif

template <int N> __device__ __forceinline__ void w4_wait_lgkmcnt() { __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8)); }

template <bool K3, int MTG>
__global__ __launch_bounds__(256, 2) void conv_planes_w4_kernel(const ConvParamsP p) {
    constexpr int NP = 2, MT = 3, NT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef YV3_TIMELINE          // measurement builds only (-DYV3_MEASURE -DYV3_TIMELINE): cycle split of one workgroup -> alpha[0..31]
    const unsigned long long tl_entry = __builtin_amdgcn_s_memtime();
    unsigned long long tl_t = tl_entry, tl_pro = 0, tl_k0 = 0, tl_wait = 0, tl_bar = 0, tl_k1 = 0, tl_epi = 0, tl_idx = 0, tl_iss = 0;
#define W4_MARK(acc_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - tl_t; tl_t = t_; } while (0)
#else
#define W4_MARK(acc_) do {} while (0)
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int bid = yv3_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % p.ntiles) * W4_BN;
    const int m0 = (bid / p.ntiles) * W4_BM;

    // ---- weight side (its addresses need no division): wave w stages weight rows [32 w, 32 w + 32) of both planes; the packed tile
    // [n/128][k/32][plane][n%128][slot][8] is already the (swizzled) LDS image, a wave instruction copies 16 rows = 1 KB linearly
    const long long btile = (long long)(n0 / 128) * p.nk;
    const int bin = (32 * wid + (lane >> 2)) * PBK + (lane & (SLOTS - 1)) * 8;
    // ---- pixel side: wave w stages tile rows [48 w, 48 w + 48) of both planes, three wave instructions per plane (lane -> row lane/4,
    // physical slot lane%4; the source is the un-swizzled slot).  48 w is a multiple of 16: the swizzle follows lane >> 4.
    const int sslot = ((lane & (SLOTS - 1)) ^ ((lane >> 4) & (SLOTS - 1))) * 8;
    long long aoff[W4_AQ];
    int ahi[W4_AQ], awi[W4_AQ];
    bool aok[W4_AQ];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int q = 0; q < W4_AQ; ++q) {
        const int m = m0 + 48 * wid + RPG * q + (lane >> 2);
        aok[q] = m < p.M;
        const int mm = aok[q] ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        if (K3) {
            ahi[q] = ho * p.stride - 1; awi[q] = wo * p.stride - 1;
            aoff[q] = (((long long)b * p.H + ahi[q]) * p.W + awi[q]) * p.Cin + sslot;
        } else {
            ahi[q] = awi[q] = 0;
            aoff[q] = (((long long)b * p.H + ho * p.stride) * p.W + wo * p.stride) * p.Cin + sslot;
        }
    }
    int kh = 0, kw = 0, c0 = 0;
    const u16* ap[W4_AQ];
    long long aps[W4_AQ];
    int ainc[W4_AQ];
    const u16* wbp = p.w;
    unsigned char* dst = lds;
    bool tapinit = true;
    auto dma_prepare = [&](int kc, int stage) {
        dst = lds + stage * W4_STAGE;
        if (c0 == 0 || tapinit) {                               // wave-uniform: first chunk of a filter tap
            tapinit = false;
#pragma unroll
            for (int q = 0; q < W4_AQ; ++q) {
                bool ok = aok[q];
                long long off = aoff[q] + c0;
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
            for (int q = 0; q < W4_AQ; ++q) ap[q] += ainc[q];
        }
        wbp = p.w + ((btile + kc) * NP) * (long long)(128 * PBK) + bin;
        c0 += PBK;
        if (c0 == p.Cin) { c0 = 0; if (++kw == 3) { kw = 0; ++kh; } }
    };
    auto dma_piece = [&](int idx) {
        if (idx < NP * W4_AQ) {
            const int q = idx / NP, pl = idx % NP;
            __builtin_amdgcn_global_load_lds(GPTR(ap[q] + pl * aps[q]), LPTR(dst + pl * W4_A_PLANE + (48 * wid + RPG * q) * ROWB), 16, 0, 0);
        } else {
            const int q = (idx - NP * W4_AQ) / NP, pl = (idx - NP * W4_AQ) % NP;
            __builtin_amdgcn_global_load_lds(GPTR(wbp + (long long)pl * (128 * PBK) + q * (RPG * PBK)),
                                             LPTR(dst + NP * W4_A_PLANE + pl * W4_B_PLANE + (32 * wid + RPG * q) * ROWB), 16, 0, 0);
        }
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    W4_MARK(tl_idx);
    // ---- ring fill: chunks 0 and 1
#pragma unroll
    for (int d = 0; d < W4_NS; ++d)
        if (d < p.nk) {
            dma_prepare(d, d);
#pragma unroll
            for (int g = 0; g < W4_G; ++g) dma_piece(g);
        }

    W4_MARK(tl_iss);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int fsw = swz(l31);
    const int x_row = (wm * 96 + l31) * ROWB;                                   // pixel fragments (B operand), block j: + j * 32 rows
    const int w_row = NP * W4_A_PLANE + (wn * 64 + l31) * ROWB;                 // weight fragments (A operand), block i: + i * 32 rows
    constexpr int NF = (NT + MT) * NP;                                          // fragments per k-step: [0, NT*NP) weights (i, plane), then pixels (j, plane)
    bf16x8v frag[2][NF];
    auto read_frag = [&](const unsigned char* st, int ks, int f) {
        const int fslot = ((ks * 2 + lhi) ^ fsw) * 16;
        if (f < NT * NP) frag[ks][f] = *reinterpret_cast<const bf16x8v*>(st + w_row + fslot + (f / NP) * 32 * ROWB + (f % NP) * W4_B_PLANE);
        else { const int g = f - NT * NP;
               frag[ks][f] = *reinterpret_cast<const bf16x8v*>(st + x_row + fslot + (g / NP) * 32 * ROWB + (g % NP) * W4_A_PLANE); }
    };

    // chunk 0 has landed and is visible; its first k-step's fragments go to registers
    if (p.nk >= 2) wait_vmcnt<W4_G>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < NF; ++f) read_frag(lds, 0, f);
    W4_MARK(tl_pro);

    // the 18 MFMAs of k-step ks (per accumulator: w_lo x_hi, w_hi x_lo, w_hi x_hi -- conv_planes_kernel's order), rotating over the six
    // accumulators; between(mi) is called after MFMA number mi
    auto kstep = [&](auto ks_c, auto&& between) {
        constexpr int ks = decltype(ks_c)::value;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int u = 0; u < NT * MT; ++u) {
                const int i = u / MT, j = u % MT;
                acc[i][j] = PlaneOps<2>::mfma(frag[ks][i * NP + (t == 0 ? 1 : 0)], frag[ks][NT * NP + j * NP + (t == 1 ? 1 : 0)], acc[i][j]);
                between(t * NT * MT + u);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    int cur = 0;
    // STEADY: chunks kc+1 and kc+2 exist (no per-piece conditions between the MFMAs); the last two chunks run the general form
    auto body = [&](int kc, auto steady_c) {
        constexpr bool STEADY = decltype(steady_c)::value;
        const unsigned char* const st = lds + cur * W4_STAGE;
        const unsigned char* const st_next = lds + (cur ^ 1) * W4_STAGE;
        const bool next = STEADY || kc + 1 < p.nk;
        const bool more = STEADY || kc + W4_NS < p.nk;
        // ---- k-step 0 from registers; behind its first MFMAs the reads of k-step 1's fragments (same stage)
        const bool wpend = YV3_W4_SPLIT_DMA && kc >= 1 && (STEADY || kc + 1 < p.nk);     // weight pieces of chunk kc+1, prepared one body ago
        kstep(I0{}, [&](int mi) {
            if (mi < NF) read_frag(st, 1, mi);
            if (wpend && mi < NP * W4_BQ) dma_piece(NP * W4_AQ + mi);
        });
        W4_MARK(tl_k0);
        // ---- the chunk's barrier: every wave has taken all it needs from this chunk's stage; my pieces of chunk kc+1 have landed
        if (next) {
            wait_vmcnt<0>();
            w4_wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            W4_MARK(tl_wait);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            W4_MARK(tl_bar);
        }
        if (more) dma_prepare(kc + W4_NS, cur);
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-step 1; behind its MFMAs the refill of this stage (chunk kc+2) and the next chunk's k-step-0 fragments.  The four waves
        // leave the barrier together: ten DMA pieces per wave behind the first MFMAs are 40 KB for the CU's one texture-address path
        // (1 KB per ~16 cycles, tools/probes/dma_rate.hip) inside ~300 cycles, and the waves' issue stalls behind it (k-step 1 took 1690
        // cycles against k-step 0's 700 with the same 18 MFMAs, profiles/r05f_w4_timeline.txt).  YV3_W4_SPLIT_DMA: only the six pixel
        // pieces go out here; the four weight pieces (L2-resident lines: short latency) follow behind the first MFMAs of the NEXT
        // chunk's k-step 0, still half a chunk before the barrier that needs them.
        kstep(I1{}, [&](int mi) {
            if (more && mi < (YV3_W4_SPLIT_DMA ? NP * W4_AQ : W4_G)) dma_piece(mi);
            if (next && mi < NF) read_frag(st_next, 0, mi);
        });
        W4_MARK(tl_k1);
        cur ^= 1;
    };
    body(0, std::false_type{});
    int kc = 1;
    for (; kc + W4_NS < p.nk; ++kc) body(kc, std::true_type{});
    for (; kc < p.nk; ++kc) body(kc, std::false_type{});

    epilogue_store<2, W4_BM, W4_BN, 2, 2, false, true, MTG>(acc, p, lds, m0, n0, wid, lane);
#ifdef YV3_TIMELINE
    W4_MARK(tl_epi);
    if (blockIdx.x == (unsigned)p.tune[2] && lane == 0 && p.alpha) {
        float* dbg = const_cast<float*>(p.alpha) + wid * 8;
        const float n_ = (float)p.nk;
        dbg[0] = (float)tl_pro; dbg[1] = tl_k0 / n_; dbg[2] = tl_wait / n_; dbg[3] = tl_bar / n_; dbg[4] = tl_k1 / n_; dbg[5] = (float)tl_epi;
        dbg[6] = n_; dbg[7] = (float)(tl_t - tl_entry);
        float* dbg2 = const_cast<float*>(p.alpha) + 32 + wid * 2;                               // prologue split: index math, DMA issue of two chunks
        dbg2[0] = (float)tl_idx; dbg2[1] = (float)tl_iss;
    }
#endif
}

}  // namespace

// 0: launched; -100: shape not taken (the caller falls through to conv_planes_kernel)
int yv3_conv2d_planes_w4(const ConvParamsP* pp, int np, int npad, hipStream_t s) {
    ConvParamsP p = *pp;
    if (np != 2 || npad % 128 || p.Cin % PBK || p.Cup > 0 || p.dec_out || p.K % PBK) return -100;
    const bool k3 = p.K == 9 * p.Cin;
    if (!k3 && p.K != p.Cin) return -100;
    p.ntiles = npad / 128;
    p.tb = 128;
    p.nk = p.K / PBK;
    if (p.nk < 2) return -100;
    const dim3 grid((unsigned)(((p.M + W4_BM - 1) / W4_BM) * p.ntiles));
    p.total = (int)grid.x;
    constexpr int MTG = 1;
    const size_t pipe = (size_t)W4_NS * W4_STAGE, epi = (size_t)4 * (MTG * 32) * (64 + 4) * 4;
    const size_t lds = pipe > epi ? pipe : epi;
    if (k3) hipLaunchKernelGGL((conv_planes_w4_kernel<true, MTG>), grid, dim3(256), lds, s, p);
    else    hipLaunchKernelGGL((conv_planes_w4_kernel<false, MTG>), grid, dim3(256), lds, s, p);
    YV3_CHECK_LAUNCH();
    return 0;
}
