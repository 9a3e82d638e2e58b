// fp16-plane convolution, 256x128 tile on FOUR waves, two workgroups per CU ("w4").
//
// Why another main loop (round 5).  conv_planes_kernel's 256x128 tile runs on eight waves with a 144 KB LDS ring, ONE
// workgroup per CU: a tile's prologue (index math + the first HBM round trip, 7-10 k cycles), its epilogue (12-19 k: residual
// rows in, two output planes out) and the gap until the next workgroup starts (~6 k) overlap nothing -- 26-43 % of a tile's time
// at 36 / 18 K chunks (profiles/r04an_persistent_stream_ab.txt); making the workgroup persistent hid the prologue but put every CU's
// HBM-heavy epilogue in phase (-0.5 %).  This kernel keeps the hardware's dynamic one-tile dispatch and instead makes TWO
// workgroups fit on a CU, so that one's prologue / epilogue / launch gap sits under the other's main loop:
//   * 16-deep K chunks (32-byte tile rows): a ring stage is 2 planes x (256 + 128) rows x 32 B = 24 KB, three stages 72 KB;
//   * four waves with 128x64 wave tiles (8 accumulator blocks = 128 registers, <= 256 per wave): per 16-deep k-step a wave
//     issues 12 ds_read_b128 for 24 MFMAs (the eight-wave tile: 8 for 12) -- a wave's LDS reads cost it ~36 cycles of issue
//     each (tools/probes/lds_read_rate.hip), the co-limiter of the eight-wave loop;
//   * single-phase rolling loop: the chunk's one barrier sits between its two halves (pixel blocks 0-1 | 2-3); after it the
//     stage just consumed is refilled by DMA three chunks ahead and the NEXT chunk's fragments are read under the second
//     half's MFMAs (weights double-buffered in registers, pixel fragments reloaded block by block as their last MFMA issues),
//     so no LDS or DMA latency is exposed behind the barrier; the other workgroup's waves fill what is.
// Same K order and the same three-product order per k-step as conv_planes_kernel: results are BIT-IDENTICAL to it
// (tools/tile_ab.py asserts that), and the packed weights are shared (a 16-deep chunk is one aligned half of a packed
// 32-deep row; the XOR swizzle of that image only permutes the two 16-byte slots inside the half).
//
// Replaces the same reference call sites as conv_planes.hip (darknet.py:43-44, :52-53): conv_bn_relu / res_layer.conv2 with
// 3x3 (stride 1 or 2) or plain 1x1 filters, plane outputs, cout_pad % 128 == 0.
#include <type_traits>
#include "conv_planes_common.h"

namespace {

constexpr int W4_BK = 16;                       // K elements per chunk
constexpr int W4_ROWB = W4_BK * 2;              // bytes per tile row per plane
constexpr int W4_BM = 256, W4_BN = 128;
constexpr int W4_NS = 3;                        // ring stages
constexpr int W4_A_PLANE = W4_BM * W4_ROWB;     // 8 KB
constexpr int W4_B_PLANE = W4_BN * W4_ROWB;     // 4 KB
constexpr int W4_STAGE = 2 * (W4_A_PLANE + W4_B_PLANE);
constexpr int W4_G = 6;                         // DMA wave instructions per chunk and wave: 2 pixel pieces x 2 planes + 1 weight piece x 2 planes

template <int N> __device__ __forceinline__ void w4_wait_lgkmcnt() { __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8)); }

// 32-byte rows read with ds_read_b128: eight consecutive rows span one 256-byte bank row, rows r and r+4 share banks -> the two
// 16-byte slots of rows 4..7 (mod 8) are swapped
__device__ __forceinline__ int w4_swz(int row) { return (row >> 2) & 1; }

template <bool K3, int MTG>
__global__ __launch_bounds__(256, 2) void conv_planes_w4_kernel(const ConvParamsP p) {
    constexpr int NP = 2, MT = 4, NT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int bid = yv3_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % p.ntiles) * W4_BN;
    const int m0 = (bid / p.ntiles) * W4_BM;

    // ---- weight side first (its addresses need no division): wave w stages weight rows [32 w, 32 w + 32) of both planes, one
    // wave instruction per plane (lane -> row lane/2, physical slot lane&1).  Source: the packed 32-deep image
    // [n/128][k/32][plane][n%128][slot ^ swz32(row)][8]; chunk kc = half (kc & 1) of packed chunk kc >> 1.
    const int brow = 32 * wid + (lane >> 1);
    const int bls = (lane & 1) ^ w4_swz(brow);                              // logical 16-byte slot this lane carries
    const int bsw = swz(brow);                                              // swizzle of the packed 64-byte row
    const int boff[2] = {brow * PBK + ((0 + bls) ^ bsw) * 8, brow * PBK + ((2 + bls) ^ bsw) * 8};
    const long long btile = (long long)(n0 / 128) * (p.nk >> 1);           // packed chunks of the preceding n-tiles
    // ---- pixel side: wave w stages tile rows [64 w, 64 w + 64) of both planes, two wave instructions per plane
    long long aoff[2];
    int ahi[2], awi[2];
    bool aok[2];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 64 * wid + 32 * q + (lane >> 1);
        const int als = ((lane & 1) ^ w4_swz(r)) * 8;
        const int m = m0 + r;
        aok[q] = m < p.M;
        const int mm = aok[q] ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        if (K3) {
            ahi[q] = ho * p.stride - 1; awi[q] = wo * p.stride - 1;
            aoff[q] = (((long long)b * p.H + ahi[q]) * p.W + awi[q]) * p.Cin + als;
        } else {
            ahi[q] = awi[q] = 0;
            aoff[q] = (((long long)b * p.H + ho * p.stride) * p.W + wo * p.stride) * p.Cin + als;
        }
    }
    int kh = 0, kw = 0, c0 = 0;
    const u16* ap[2];
    long long aps[2];
    int ainc[2];
    const u16* wbp = p.w;
    unsigned char* dst = lds;
    bool tapinit = true;
    auto dma_prepare = [&](int kc, int stage) {
        dst = lds + stage * W4_STAGE;
        if (c0 == 0 || tapinit) {                               // wave-uniform: first chunk of a filter tap
            tapinit = false;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bool ok = aok[q];
                long long off = aoff[q] + c0;
                if (K3) {
                    ok = ok && (unsigned)(ahi[q] + kh) < (unsigned)p.H && (unsigned)(awi[q] + kw) < (unsigned)p.W;
                    off += ((long long)kh * p.W + kw) * p.Cin;
                }
                ap[q] = ok ? p.x + off : g_zero_page;
                aps[q] = ok ? p.xs : 0;
                ainc[q] = ok ? W4_BK : 0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) ap[q] += ainc[q];
        }
        wbp = p.w + ((btile + (kc >> 1)) * NP) * (long long)(128 * PBK) + boff[kc & 1];
        c0 += W4_BK;
        if (c0 == p.Cin) { c0 = 0; if (++kw == 3) { kw = 0; ++kh; } }
    };
    auto dma_piece = [&](int idx) {
        if (idx < 4) {
            const int q = idx >> 1, pl = idx & 1;
            __builtin_amdgcn_global_load_lds(GPTR(ap[q] + pl * aps[q]), LPTR(dst + pl * W4_A_PLANE + (64 * wid + 32 * q) * W4_ROWB), 16, 0, 0);
        } else {
            const int pl = idx - 4;
            __builtin_amdgcn_global_load_lds(GPTR(wbp + (long long)pl * (128 * PBK)), LPTR(dst + NP * W4_A_PLANE + pl * W4_B_PLANE + (32 * wid) * W4_ROWB), 16, 0, 0);
        }
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- ring fill: chunks 0 .. 2
#pragma unroll
    for (int d = 0; d < W4_NS; ++d)
        if (d < p.nk) {
            dma_prepare(d, d);
#pragma unroll
            for (int g = 0; g < W4_G; ++g) dma_piece(g);
        }

    const int l31 = lane & 31, lhi = lane >> 5;
    const int fslot = (lhi ^ w4_swz(l31)) * 16;
    const int x_row = (wm * 128 + l31) * W4_ROWB + fslot;                          // pixel fragments (B operand), block j: + j * 32 rows
    const int w_row = NP * W4_A_PLANE + (wn * 64 + l31) * W4_ROWB + fslot;         // weight fragments (A operand), block i: + i * 32 rows
    // fragments in registers: pixel blocks 0..2 single-buffered (block j of the next chunk is read once block j's last MFMA of this
    // chunk has issued), the LAST block double-buffered like the weights (its reload would otherwise be the only read still in flight at
    // the top of the next chunk, in front of the wait for the first MFMA's operands)
    bf16x8v xf[MT - 1][NP], xl[2][NP], wf[2][NT][NP];
    auto read_x = [&](const unsigned char* st, int j) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) xf[j][pl] = *reinterpret_cast<const bf16x8v*>(st + x_row + j * 32 * W4_ROWB + pl * W4_A_PLANE);
    };
    auto read_xl = [&](const unsigned char* st, int buf) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) xl[buf][pl] = *reinterpret_cast<const bf16x8v*>(st + x_row + (MT - 1) * 32 * W4_ROWB + pl * W4_A_PLANE);
    };
    auto read_w = [&](const unsigned char* st, int buf, int i) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) wf[buf][i][pl] = *reinterpret_cast<const bf16x8v*>(st + w_row + i * 32 * W4_ROWB + pl * W4_B_PLANE);
    };

    // chunk 0 has landed and is visible; its fragments go to registers
    if (p.nk >= 3) wait_vmcnt<2 * W4_G>(); else if (p.nk == 2) wait_vmcnt<W4_G>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < NT; ++i) read_w(lds, 0, i);
#pragma unroll
    for (int j = 0; j < MT - 1; ++j) read_x(lds, j);
    read_xl(lds, 0);

    // the six MFMAs of pixel block j (per accumulator: w_lo x_hi, w_hi x_lo, w_hi x_hi -- conv_planes_kernel's order)
    auto mfma_block = [&](int buf, int j, auto&& between) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const bf16x8v xv = j == MT - 1 ? xl[buf][t == 1 ? 1 : 0] : xf[j < MT - 1 ? j : 0][t == 1 ? 1 : 0];
                acc[i][j] = PlaneOps<2>::mfma(wf[buf][i][t == 0 ? 1 : 0], xv, acc[i][j]);
                between(t * NT + i);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto nothing = [](int) {};

    int cur = 0;
    // STEADY: chunks kc+1 .. kc+3 all exist (no per-piece conditions between the MFMAs); the last three chunks run the general form
    auto body = [&](int kc, auto buf_c, auto steady_c) {
        constexpr int buf = decltype(buf_c)::value;
        constexpr bool STEADY = decltype(steady_c)::value;
        const unsigned char* const st_next = lds + (cur + 1 == W4_NS ? 0 : cur + 1) * W4_STAGE;
        const bool next = STEADY || kc + 1 < p.nk;
        const bool more = STEADY || kc + W4_NS < p.nk;
        // ---- first half: pixel blocks 0 and 1 from registers
        mfma_block(buf, 0, nothing);
        mfma_block(buf, 1, nothing);
        // ---- the chunk's barrier: my pieces of chunk kc+1 have landed (chunk kc+2's may stay in flight) and every wave is done
        // with this chunk's stage (all of its fragments went to registers before the first half)
        if (next) {
            if (STEADY || kc + 2 < p.nk) wait_vmcnt<W4_G>(); else wait_vmcnt<0>();
            w4_wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) dma_prepare(kc + W4_NS, cur);
        __builtin_amdgcn_sched_barrier(0);
        // ---- second half: pixel blocks 2 and 3; behind the MFMAs the refill of this stage (chunk kc+3) and the next chunk's fragments
        mfma_block(buf, 2, [&](int mi) {
            if (next) {
                if (mi == 0) read_w(st_next, buf ^ 1, 0);
                if (mi == 1) read_w(st_next, buf ^ 1, 1);
                if (mi == 2) read_x(st_next, 0);
                if (mi == 3) read_x(st_next, 1);
            }
            if (more) dma_piece(mi);
        });
        mfma_block(buf, 3, [&](int mi) {
            if (next && mi == 0) { read_x(st_next, 2); read_xl(st_next, buf ^ 1); }
        });
        cur = cur + 1 == W4_NS ? 0 : cur + 1;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    int kc = 0;
    for (; kc + 1 + W4_NS < p.nk; kc += 2) {                     // pairs (register double buffer of the weight fragments)
        body(kc, I0{}, std::true_type{});
        body(kc + 1, I1{}, std::true_type{});
    }
    for (; kc + 1 < p.nk; kc += 2) {
        body(kc, I0{}, std::false_type{});
        body(kc + 1, I1{}, std::false_type{});
    }
    if (kc < p.nk) body(kc, I0{}, std::false_type{});

    epilogue_store<2, W4_BM, W4_BN, 2, 2, false, true, MTG>(acc, p, lds, m0, n0, wid, lane);
}

}  // namespace

// 1: launched; -100: shape not taken (the caller falls through to conv_planes_kernel)
int yv3_conv2d_planes_w4(const ConvParamsP* pp, int np, int npad, hipStream_t s) {
    ConvParamsP p = *pp;
    if (np != 2 || npad % 128 || p.Cin % W4_BK || p.Cup > 0 || p.dec_out || p.K % PBK) return -100;
    const bool k3 = p.K == 9 * p.Cin;
    if (!k3 && p.K != p.Cin) return -100;
    p.ntiles = npad / 128;
    p.tb = 128;
    p.nk = p.K / W4_BK;
    if (p.nk < 3) return -100;
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
