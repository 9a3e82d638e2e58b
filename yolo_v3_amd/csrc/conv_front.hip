// The first two layers of Darknet-53 as ONE kernel (fp16x2-plane mode):
//     feature.mlist.0  conv_bn_relu(3 -> 32, 3x3, s1)   reference darknet.py:76, :43-44
//     feature.mlist.1  conv_bn_relu(32 -> 64, 3x3, s2)  reference darknet.py:68-70 (make_res_stack's down-sampling conv)
//
// Unfused they are HBM-bound: the first layer writes a [B,H,W,32] activation (1.42 GB at 416x416 bs=64, 4 B per element in
// this mode) that the second immediately re-reads -- 1.07 ms of a 14.9 ms step for 0.12 TFLOP of work.  Here the first
// layer's output never leaves the CU: a persistent workgroup walks 8x16-pixel tiles of the SECOND layer's output and for
// each tile
//   1. stages the 19 x 35 x 3 input patch (NCHW fp32, zero halo, x16) in LDS           (prefetched one tile ahead in registers)
//   2. computes the 17 x 33 first-layer pixels the tile needs on the matrix cores (K = 27 padded to 32, operands split hi+lo
//      in registers: the arithmetic of conv0.hip's conv0_mfma_kernel, instruction for instruction) and writes them, BN +
//      LeakyReLU applied and split into fp16 hi/lo planes, into an LDS-resident image -- zero where the pixel lies outside the
//      image (the second conv's padding).  The image is stored by column parity, [parity][17 rows][18], so that the stride-2
//      taps of the second conv read CONSECUTIVE 64-byte rows, XOR-swizzled like every other tile: conflict-free
//      ds_read_b128 for all nine taps (checked exhaustively against the b128 lane groups);
//   3. runs the second conv (M = 128 pixels, N = 64, K = 9 taps x 32) entirely out of LDS: its packed weights (72 KB) are
//      DMA-ed once per workgroup and stay resident, so the main loop has no global traffic and no barrier;
//   4. BN + LeakyReLU + hi/lo split + full-line stores, as conv_planes_common.h's epilogue.
// Same products in the same order as yv3_conv0 followed by yv3_conv2d on the cout-64 tile configuration: the results are
// bit-identical to the two-launch path (tests/test_gpu_kernels.py::test_fused_front_equals_two_launches_bitwise).
// HBM traffic: 12 B in + 256 B out per second-layer pixel (0.84 GB per step instead of 3.7 GB).
// LDS: 78 336 (image, re-used by the epilogue transposes) + 73 728 (weights) + 8 208 (patch) = 160 272 B: one workgroup per CU.
#include <type_traits>
#include "conv_planes_common.h"

namespace {

constexpr int FT_R = 8, FT_C = 16;                        // output tile of the second conv (rows x cols)
constexpr int FR_COLS = 2 * FT_C + 1;                     // first-layer region: 17 rows x 33 cols
constexpr int FR_PX = (2 * FT_R + 1) * FR_COLS;           // 561
constexpr int FR_GROUPS = (FR_PX + 31) / 32;              // 18 groups of 32 pixels
constexpr int FP_ROWS = 2 * FT_R + 3, FP_COLS = 2 * FT_C + 3;   // input patch 19 x 35
constexpr int FP_PITCH = 36, FP_CH = FP_ROWS * FP_PITCH;  // floats
constexpr int FA_RP = 18, FA_PB = (2 * FT_R + 1) * FA_RP; // image row pitch (pixels), parity block (306 pixels)
constexpr int FA_PLANE = 2 * FA_PB * ROWB;                // 39 168 bytes per plane
constexpr int F_A_OFF = 0;
// NP = planes of the second layer's operands / of the output: 2 = fp16 hi + lo (YV3_F32_F16X2), 1 = bf16 (YV3_BF16: the first
// layer is computed exactly as in the fp16-plane mode -- conv0.hip's conv0_mfma_kernel<true> -- and rounded to bf16)
template <int NP> struct FrontGeo {
    static constexpr int FW_CHUNK = NP * 64 * ROWB;           // packed weights per tap: NP planes x 64 rows x 64 B
    static constexpr int FW_BYTES = 9 * FW_CHUNK;             // 73 728 (NP = 2)
    static constexpr int F_W_OFF = NP * FA_PLANE, F_P_OFF = F_W_OFF + FW_BYTES;
    static constexpr int F_LDS = F_P_OFF + 3 * FP_CH * 4;     // 160 272 (NP = 2), 84 240 (NP = 1)
};
constexpr int F_EP = 32 + 4;                              // floats per row of a wave's epilogue transpose tile

typedef _Float16 fh16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fh16x2 __attribute__((ext_vector_type(2)));
typedef float ff32x2 __attribute__((ext_vector_type(2)));

__device__ inline void fsplit8(const float (&v)[8], fh16x8& hi, fh16x8& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const ff32x2 a = {v[2 * q], v[2 * q + 1]};
        const fh16x2 h = __builtin_convertvector(a, fh16x2);
        const ff32x2 r = a - __builtin_convertvector(h, ff32x2);
        const fh16x2 l = __builtin_convertvector(r, fh16x2);
        hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
    }
}

struct FrontParams {
    const float* x;          // [B,3,H,W] fp32
    const float* w0;         // first layer weights [27][32] fp32 (tap-major)
    const float* alpha0; const float* beta0;
    const u16* w1;           // second layer: packed fp16x2 planes (yv3_pack_conv_weight, cout_pad 64)
    const float* alpha1; const float* beta1;
    u16* y;                  // [2][B,H/2,W/2,64]
    long long ys;            // plane stride of y (elements)
    int H, W, B;
    int tiles_x, tiles_y, total;
    int* flags;
};

template <int NP>
__global__ __launch_bounds__(512) void conv_front_kernel(const FrontParams p) {
    constexpr int FW_CHUNK = FrontGeo<NP>::FW_CHUNK, FW_BYTES = FrontGeo<NP>::FW_BYTES, F_W_OFF = FrontGeo<NP>::F_W_OFF,
                  F_P_OFF = FrontGeo<NP>::F_P_OFF;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned* const patch = reinterpret_cast<unsigned*>(lds + F_P_OFF);      // each element: fp16 hi | fp16 lo << 16 of 16 * x
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);                // 8 waves: two per SIMD
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;                // wave tile: 32 pixels (tile rows 2wm, 2wm+1) x 32 channels
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    bool bad = false;

    // ---- second-layer weights: one linear DMA of the packed array (already in LDS-image order), resident for the launch
    for (int pc = wid; pc < FW_BYTES / 1024; pc += 8)
        __builtin_amdgcn_global_load_lds(GPTR(p.w1 + pc * 512 + lane * 8), LPTR(lds + F_W_OFF + pc * 1024), 16, 0, 0);

    // ---- first-layer A operand (weights), once per wave: exactly conv0_mfma_kernel's assignment (conv0.hip)
    const int we = (l31 & 3) + 4 * (l31 >> 3), wh = (l31 >> 2) & 1;
    const int ch0 = (we & 7) + 8 * wh + 16 * (we >> 3);
    fh16x8 whi[2], wlo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = ks == 0 ? lhi * 9 + j : (lhi == 0 ? 18 + j : (j < 3 ? j * 9 + 8 : -1));
            wv[j] = t >= 0 ? p.w0[t * 32 + ch0] * 256.f : 0.f;
            bad |= !(__builtin_fabsf(wv[j]) <= 65504.f);
        }
        fsplit8(wv, whi[ks], wlo[ks]);
    }
    float al0[16], be0[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { const int c = (e & 7) + 8 * lhi + 16 * (e >> 3); al0[e] = p.alpha0[c] * (1.f / 4096.f); be0[e] = p.beta0[c]; }
    int d1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        d1[j] = lhi == 0 ? 2 * FP_CH + (j / 3) * FP_PITCH + j % 3 : (j % 3) * FP_CH + 2 * FP_PITCH + 2;

    // ---- second-layer epilogue constants: BN scale / shift of this lane's channels (4 groups of 4)
    f32x4 alv[4], bev[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = wn * 32 + 8 * g + 4 * lhi;
        alv[g] = *reinterpret_cast<const f32x4*>(p.alpha1 + n);
        bev[g] = *reinterpret_cast<const f32x4*>(p.beta1 + n);
    }

    // ---- fragment addresses of the second conv (constant over tiles).  Pixel side: lane -> (tile row, tile col) of its
    // output pixel; image pixel of tap (kh,kw): parity = kw&1, row 2r+kh, half-column c + (kw>>1).
    const int pr = l31 >> 4, pcx = l31 & 15;
    const int pbase = (2 * (2 * wm + pr)) * FA_RP + pcx;                          // kh = kw = 0
    int xa[2][2];                                                                 // [ks][kw == 2]: byte offset incl. swizzled slot
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        xa[ks][0] = pbase * ROWB + (((ks * 2 + lhi) ^ ((pcx >> 2) & 3)) * 16);
        xa[ks][1] = (pbase + 1) * ROWB + (((ks * 2 + lhi) ^ (((pcx + 1) >> 2) & 3)) * 16);
    }
    int wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wa[ks] = F_W_OFF + (wn * 32 + l31) * ROWB + (((ks * 2 + lhi) ^ swz(l31)) * 16);

    // ---- first-layer pixel groups of this wave (g = wid, wid+8, wid+16): patch base, image address, position -- tile-independent
    constexpr int GW = (FR_GROUPS + 7) / 8;               // 3
    int gbase[GW], gimg[GW], grow[GW], gcol[GW];
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        const int idx = (wid + 8 * i) * 32 + l31;
        const bool live = idx < FR_PX;
        const int ii = live ? idx : 0;
        const int row = ii / FR_COLS, col = ii - row * FR_COLS;
        gbase[i] = row * FP_PITCH + col;
        const int colh = col >> 1;
        const int pi = (col & 1) * FA_PB + row * FA_RP + colh;
        gimg[i] = live ? F_A_OFF + pi * ROWB + (((colh >> 2) & 3) << 20) : -1;      // bits 20.. : swizzle; -1: lane has no pixel
        grow[i] = row; gcol[i] = col;
    }

    // ---- patch elements of this thread (i = tid + 512*k of 3 x 19 x 35): LDS index, offset in the image, position -- tile-independent
    constexpr int PN = 3 * FP_ROWS * FP_COLS;             // 1995
    constexpr int PK = (PN + 511) / 512;                  // 4
    int plds[PK], prr[PK], pcc[PK];
    long long pgo[PK];
#pragma unroll
    for (int k = 0; k < PK; ++k) {
        const int i = tid + 512 * k;
        const int c = i / (FP_ROWS * FP_COLS);
        const int r2 = i - c * (FP_ROWS * FP_COLS);
        prr[k] = r2 / FP_COLS; pcc[k] = r2 - prr[k] * FP_COLS;
        plds[k] = i < PN ? c * FP_CH + prr[k] * FP_PITCH + pcc[k] : -1;
        pgo[k] = ((long long)c * p.H + prr[k] - 2) * p.W + pcc[k] - 2;
    }
    float pre[PK];
    auto patch_fetch = [&](int tile) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const float* xb = p.x + (size_t)b * 3 * p.H * p.W + (long long)(2 * FT_R * ty) * p.W + 2 * FT_C * tx;
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            const int gy = 2 * FT_R * ty - 2 + prr[k], gx = 2 * FT_C * tx - 2 + pcc[k];
            float v = 0.f;
            if (plds[k] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = xb[pgo[k]];
            pre[k] = v;
        }
    };
    if ((int)blockIdx.x < p.total) patch_fetch(blockIdx.x);

    float amax = 0.f;
#ifdef YV3_FRONT_TL      // debug build only (tools/timeline.py --kernel front): cycle split of one workgroup, written over y[0..]
    unsigned long long tl_t = __builtin_amdgcn_s_memtime(), tl_acc[6] = {0, 0, 0, 0, 0, 0};
#define FTL(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_acc[i] += t_ - tl_t; tl_t = t_; } while (0)
#else
#define FTL(i) do {} while (0)
#endif
    for (int tile = blockIdx.x; tile < p.total; tile += gridDim.x) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int r0 = FT_R * ty, c0 = FT_C * tx;

        // ---- 1. patch -> LDS: 16 * x (exact; undone in alpha) split ONCE per element into fp16 hi | lo (the split conv0.hip
        // performs per use gives the same halves).  Every wave is past the previous tile's epilogue here.
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            if (plds[k] >= 0) {
                bad |= !(__builtin_fabsf(pre[k]) <= 4094.f);
                const float a = pre[k] * 16.f;
                const _Float16 h = (_Float16)a;
                const _Float16 l = (_Float16)(a - (float)h);
                patch[plds[k]] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
            }
        }
        FTL(0);
        __syncthreads();
        FTL(1);
        if (tile + (int)gridDim.x < p.total) patch_fetch(tile + gridDim.x);       // lands during steps 2-4

        // ---- 2. first layer for the 561 region pixels, 32 per MFMA column block.  A wave's first two groups are processed
        // TOGETHER (two independent MFMA chains / LDS round trips interleave instead of one dependent chain at a time)
        auto first_layer = [&](auto ntag, int i0) {
            constexpr int NG = decltype(ntag)::value;
            unsigned q0[NG][8], q1[NG][8];
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                const unsigned* p0 = patch + gbase[i0 + n] + lhi * FP_CH;
#pragma unroll
                for (int j = 0; j < 8; ++j) { q0[n][j] = p0[(j / 3) * FP_PITCH + j % 3]; q1[n][j] = patch[gbase[i0 + n] + d1[j]]; }
            }
            f32x16 acc[NG];
#pragma unroll
            for (int n = 0; n < NG; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
            u32x4 xh[NG][2], xl[NG][2];                                           // 8 fp16 each: hi / lo parts of a k-step's 8 taps
#pragma unroll
            for (int n = 0; n < NG; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xh[n][0][q] = __builtin_amdgcn_perm(q0[n][2 * q + 1], q0[n][2 * q], 0x05040100u);
                    xl[n][0][q] = __builtin_amdgcn_perm(q0[n][2 * q + 1], q0[n][2 * q], 0x07060302u);
                    xh[n][1][q] = __builtin_amdgcn_perm(q1[n][2 * q + 1], q1[n][2 * q], 0x05040100u);
                    xl[n][1][q] = __builtin_amdgcn_perm(q1[n][2 * q + 1], q1[n][2 * q], 0x07060302u);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 3; ++t)                                       // per accumulator: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi (conv0.hip's order)
#pragma unroll
                    for (int n = 0; n < NG; ++n) {
                        const fh16x8 xhi = __builtin_bit_cast(fh16x8, xh[n][ks]), xlo = __builtin_bit_cast(fh16x8, xl[n][ks]);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t == 0 ? wlo[ks] : whi[ks], t == 1 ? xlo : xhi, acc[n], 0, 0, 0);
                    }
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                const int i = i0 + n;
                const int gy = 2 * r0 - 1 + grow[i], gx = 2 * c0 - 1 + gcol[i];
                const bool inimg = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                u32x4 qh[2], ql[2];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float t0 = fmaf(acc[n][2 * q], al0[2 * q], be0[2 * q]), t1 = fmaf(acc[n][2 * q + 1], al0[2 * q + 1], be0[2 * q + 1]);
                    t0 = __builtin_fmaxf(t0, 0.1f * t0); t1 = __builtin_fmaxf(t1, 0.1f * t1);      // LeakyReLU(0.1)
                    if constexpr (NP == 2) {
                        const ff32x2 a = {t0, t1};
                        const fh16x2 h = __builtin_convertvector(a, fh16x2);
                        const fh16x2 l = __builtin_convertvector(a - __builtin_convertvector(h, ff32x2), fh16x2);
                        qh[q >> 2][q & 3] = inimg ? __builtin_bit_cast(unsigned, h) : 0u;  // outside the image: the second conv's zero padding
                        ql[q >> 2][q & 3] = inimg ? __builtin_bit_cast(unsigned, l) : 0u;
                    } else {                                                               // bf16, round to nearest even (as conv0.hip)
                        qh[q >> 2][q & 3] = inimg ? yv3_pack_bf16x2(t0, t1) : 0u;
                    }
                }
                if (gimg[i] >= 0) {
                    const int sw = gimg[i] >> 20;
                    unsigned char* d = lds + (gimg[i] & 0xfffff);
                    *reinterpret_cast<u32x4*>(d + ((lhi ^ sw) * 16)) = qh[0];                   // channels 8*lhi .. +7
                    *reinterpret_cast<u32x4*>(d + (((2 + lhi) ^ sw) * 16)) = qh[1];             // channels 16 + 8*lhi .. +7
                    if constexpr (NP == 2) {
                        *reinterpret_cast<u32x4*>(d + FA_PLANE + ((lhi ^ sw) * 16)) = ql[0];
                        *reinterpret_cast<u32x4*>(d + FA_PLANE + (((2 + lhi) ^ sw) * 16)) = ql[1];
                    }
                }
            }
        };
#if !defined(YV3_FRONT_NG) || YV3_FRONT_NG == 2
        first_layer(std::integral_constant<int, 2>{}, 0);                          // groups wid, wid + 8 (both < 18 for every wave)
        if (wid + 16 < FR_GROUPS) first_layer(std::integral_constant<int, 1>{}, 2);   // waves 0, 1: group wid + 16
#else
        first_layer(std::integral_constant<int, 1>{}, 0);
        first_layer(std::integral_constant<int, 1>{}, 1);
        if (wid + 16 < FR_GROUPS) first_layer(std::integral_constant<int, 1>{}, 2);
#endif
        // (first tile: every wave has consumed its patch registers, loaded AFTER its share of the weight DMA was issued,
        // so that DMA has landed -- vector-memory loads complete in order)
        FTL(2);
        __syncthreads();
        FTL(1);

        // ---- 3. second conv out of LDS: 9 taps x 2 k-steps, K order (kh, kw, c) as the packed weights
        f32x16 acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            const int aoff = F_A_OFF + (kh * FA_RP + (kw & 1) * FA_PB) * ROWB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8v wf[NP], xf[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    wf[pl] = *reinterpret_cast<const bf16x8v*>(lds + wa[ks] + tap * FW_CHUNK + pl * (64 * ROWB));
                    xf[pl] = *reinterpret_cast<const bf16x8v*>(lds + aoff + xa[ks][kw >> 1] + pl * FA_PLANE);
                }
                acc2 = mfma_unit<NP>(wf, xf, acc2);
            }
        }
        FTL(3);
        __syncthreads();                                                              // the image is dead: its LDS becomes the transpose tiles
        FTL(1);

        // ---- 4. epilogue: BN + LeakyReLU -> per-wave LDS transpose -> hi/lo planes, full 128-byte rows per pixel
        float* tl = reinterpret_cast<float*>(lds) + wid * (32 * F_EP);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = fmaf(acc2[4 * g + q], alv[g][q], bev[g][q]);
                v[q] = __builtin_fmaxf(t, 0.1f * t);
            }
            *reinterpret_cast<f32x4*>(tl + l31 * F_EP + 8 * g + 4 * lhi) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = ps * 16 + (lane >> 2), cg = (lane & 3) * 8;
            const int t = wm * 32 + r;                                                // pixel of the 8 x 16 tile
            const long long m = ((long long)b * Ho + r0 + (t >> 4)) * Wo + c0 + (t & 15);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(tl + r * F_EP + cg);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(tl + r * F_EP + cg + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            u16* yo = p.y + m * 64 + wn * 32 + cg;
            if constexpr (NP == 2) {
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
                *reinterpret_cast<u32x4*>(yo) = qh;
                *reinterpret_cast<u32x4*>(yo + p.ys) = ql;
            } else {                                                    // one bf16 plane, as conv_planes_common.h's epilogue (PlaneOps<1>::pack2)
                u32x4 qb;
#pragma unroll
                for (int h = 0; h < 4; ++h) qb[h] = PlaneOps<1>::pack2(v[2 * h], v[2 * h + 1]);
                *reinterpret_cast<u32x4*>(yo) = qb;
            }
        }
        FTL(4);
    }
#ifdef YV3_FRONT_TL
    if (blockIdx.x == 17 && lane == 0) {
        float* dbg = reinterpret_cast<float*>(p.y) + wid * 8;
        for (int i = 0; i < 5; ++i) dbg[i] = (float)tl_acc[i];
        dbg[5] = (float)((p.total - 17 + gridDim.x - 1) / gridDim.x);
    }
#endif
    if (p.flags && __any(bad || !(amax <= 65504.f)) && lane == 0) atomicOr(p.flags, 1);
}

}  // namespace

template <int NP>
static int conv_front_launch(const float* x_nchw, const float* w0_tap_major, const float* alpha0, const float* beta0,
                             const void* w1_packed, const float* alpha1, const float* beta1, void* y,
                             int B, int H, int W, int* flags, void* stream) {
    if (!x_nchw || !w0_tap_major || !alpha0 || !beta0 || !w1_packed || !alpha1 || !beta1 || !y || B <= 0 || H <= 0 || W <= 0)
        return YV3_EINVAL;
    if ((H % (2 * FT_R)) || (W % (2 * FT_C))) return YV3_ESHAPE;          // whole 8 x 16 output tiles only (network inputs are multiples of 32)
    FrontParams p;
    p.x = x_nchw; p.w0 = w0_tap_major; p.alpha0 = alpha0; p.beta0 = beta0;
    p.w1 = (const u16*)w1_packed; p.alpha1 = alpha1; p.beta1 = beta1; p.y = (u16*)y;
    p.H = H; p.W = W; p.B = B;
    p.ys = (long long)B * (H / 2) * (W / 2) * 64;
    p.tiles_x = (W / 2) / FT_C; p.tiles_y = (H / 2) / FT_R;
    const long long total = (long long)B * p.tiles_x * p.tiles_y;
    if (total > 0x7fffffffLL) return YV3_ESHAPE;
    p.total = (int)total;
    p.flags = flags;
    static_assert(FrontGeo<NP>::F_LDS <= 160 * 1024, "LDS budget");
    const int ncu = yv3_num_cu();
    const int grid = p.total < ncu ? p.total : ncu;                      // persistent: one workgroup per CU
    hipLaunchKernelGGL(conv_front_kernel<NP>, dim3(grid), dim3(512), FrontGeo<NP>::F_LDS, (hipStream_t)stream, p);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_conv_front(const float* x_nchw, const float* w0_tap_major, const float* alpha0, const float* beta0,
                              const void* w1_packed, const float* alpha1, const float* beta1, void* y,
                              int B, int H, int W, int* flags, void* stream) {
    return conv_front_launch<2>(x_nchw, w0_tap_major, alpha0, beta0, w1_packed, alpha1, beta1, y, B, H, W, flags, stream);
}

extern "C" int yv3_conv_front_bf16(const float* x_nchw, const float* w0_tap_major, const float* alpha0, const float* beta0,
                                   const void* w1_packed, const float* alpha1, const float* beta1, void* y,
                                   int B, int H, int W, int* flags, void* stream) {
    return conv_front_launch<1>(x_nchw, w0_tap_major, alpha0, beta0, w1_packed, alpha1, beta1, y, B, H, W, flags, stream);
}
