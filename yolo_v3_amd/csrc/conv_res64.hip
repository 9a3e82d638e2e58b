// The first residual block of Darknet-53 as ONE kernel (fp16x2-plane mode):
//     feature.mlist.2 = res_layer(64):  x + conv_bn_relu(32 -> 64, 3x3)(conv_bn_relu(64 -> 32, 1x1)(x))    reference darknet.py:46-53
// at the highest resolution after the first down-sampling (208 x 208 for a 416 input).  Unfused, the two launches are
// HBM-bound (4 B per activation element): the 1x1 reads x and writes a 32-channel map the 3x3 reads back, and the 3x3 reads
// x again as the residual: 2.83 GB per step at bs=64.  Here a persistent workgroup per CU walks 8x16-pixel output tiles:
//   1. the 10 x 18 x 64-channel region of x the tile needs is DMA-ed into LDS (global_load_lds, issued a tile ahead);
//   2. the 1x1 conv runs on the matrix cores for those 180 pixels (its weights live in registers), BN + LeakyReLU, split into
//      fp16 hi/lo and written to an LDS-resident image (zero outside the picture: the 3x3 conv's padding), pitch 20 pixels,
//      XOR-swizzled by column: conflict-free ds_read_b128 for all nine taps (checked exhaustively);
//   3. the 3x3 conv (M = 128 pixels, N = 64, K = 9 x 32) runs entirely out of LDS -- its packed weights (72 KB) are resident
//      for the whole launch -- no global traffic, no barrier inside;
//   4. epilogue: BN + LeakyReLU + residual (x rows re-read from L2, requested before step 3) + hi/lo split + full-line stores.
// Same products in the same order as yv3_conv2d (1x1) followed by yv3_conv2d (3x3 + residual): bit-identical results
// (tests/test_gpu_kernels.py::test_fused_res64_equals_two_launches_bitwise).  HBM traffic: x once (+ 41 % halo) + y once.
#include "conv_planes_common.h"

namespace {

constexpr int RT_R = 8, RT_C = 16;                        // output tile (rows x cols)
constexpr int RR_COLS = RT_C + 2, RR_PX = (RT_R + 2) * RR_COLS;   // region 10 x 18 = 180 pixels
constexpr int RR_ROWS32 = 192;                            // padded to 6 MFMA column blocks
constexpr int RX_PLANE = RR_ROWS32 * ROWB;                // x region: [2 chunks][NP planes][192 rows][64 B]
constexpr int RI_RP = 20;                                 // image pitch (pixels per row)
constexpr int RI_PLANE = (RT_R + 2) * RI_RP * ROWB;       // 12 800
constexpr int RE_BYTES = 8 * 32 * 36 * 4;                 // epilogue transposes (re-use the image's space + slack): 36 864
// NP = planes of the tensors: 2 = fp16 hi + lo (YV3_F32_F16X2), 1 = bf16 (YV3_BF16)
template <int NP> struct ResGeo {
    static constexpr int RX_CHUNK = NP * RX_PLANE, RX_BYTES = 2 * RX_CHUNK;      // 49 152 (NP = 2)
    static constexpr int RW_BYTES = 9 * NP * 64 * ROWB;                          // 73 728 (NP = 2)
    static constexpr int R_X_OFF = 0, R_I_OFF = RX_BYTES, R_W_OFF = R_I_OFF + RE_BYTES, R_LDS = R_W_OFF + RW_BYTES;   // 159 744 (NP = 2)
    static_assert(NP * RI_PLANE <= RE_BYTES && R_LDS <= 160 * 1024, "LDS budget");
};
constexpr int R_EP = 36;

struct Res64Params {
    const u16* x;            // [2][B,H,W,64] input planes (also the residual)
    const u16* w1; const float* alpha1; const float* beta1;      // 1x1 64 -> 32, packed (cout_pad 32)
    const u16* w2; const float* alpha2; const float* beta2;      // 3x3 32 -> 64, packed (cout_pad 64)
    u16* y;                  // [2][B,H,W,64]
    long long ps;            // plane stride of x and y (elements)
    int H, W, B, tiles_x, tiles_y, total;
    int* flags;
};

template <int NP>
__global__ __launch_bounds__(512) void conv_res64_kernel(const Res64Params p) {
    constexpr int RX_CHUNK = ResGeo<NP>::RX_CHUNK, RW_BYTES = ResGeo<NP>::RW_BYTES, R_X_OFF = ResGeo<NP>::R_X_OFF,
                  R_I_OFF = ResGeo<NP>::R_I_OFF, R_W_OFF = ResGeo<NP>::R_W_OFF;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;                // 3x3 wave tile: 32 pixels (tile rows 2wm, 2wm+1) x 32 channels

    // ---- 3x3 weights: resident in LDS for the launch
    for (int pc = wid; pc < RW_BYTES / 1024; pc += 8)
        __builtin_amdgcn_global_load_lds(GPTR(p.w2 + pc * 512 + lane * 8), LPTR(lds + R_W_OFF + pc * 1024), 16, 0, 0);
    // ---- 1x1 weights: this lane's fragments, resident in registers ([chunk][plane][row 32][slot ^ swz][8] packed image)
    bf16x8v wf1[2][2][NP];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                wf1[kc][ks][pl] = *reinterpret_cast<const bf16x8v*>(p.w1 + ((kc * NP + pl) * 32 + l31) * PBK + (((ks * 2 + lhi) ^ swz(l31)) * 8));
    f32x4 alv1[4], bev1[4], alv2[4], bev2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        alv1[g] = *reinterpret_cast<const f32x4*>(p.alpha1 + 8 * g + 4 * lhi);
        bev1[g] = *reinterpret_cast<const f32x4*>(p.beta1 + 8 * g + 4 * lhi);
        alv2[g] = *reinterpret_cast<const f32x4*>(p.alpha2 + wn * 32 + 8 * g + 4 * lhi);
        bev2[g] = *reinterpret_cast<const f32x4*>(p.beta2 + wn * 32 + 8 * g + 4 * lhi);
    }

    // ---- x-region DMA: 24 NP wave instructions per tile (2 chunks x NP planes x 12 groups of 16 rows); this wave issues
    // i = wid + 8k.  Per lane: region pixel of its row, source swizzle (undone by the fragment reads), all tile-independent.
    constexpr int DK = 3 * NP;                            // instructions per wave
    const int sslot = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    int drr[DK], dcc[DK];
    long long dsrc[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) {
        const int i = wid + 8 * k;
        const int kc = i / (12 * NP), pl = (i / 12) % NP, grp = i % 12;
        const int row = grp * 16 + (lane >> 2);
        drr[k] = row < RR_PX ? row / RR_COLS : -100;      // rows 180..191: never inside the picture -> zero page
        dcc[k] = row - (row / RR_COLS) * RR_COLS;
        dsrc[k] = (long long)pl * p.ps + kc * PBK + sslot;
    }
    auto x_dma = [&](int tile) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            const int i = wid + 8 * k;
            const int kc = i / (12 * NP), pl = (i / 12) % NP, grp = i % 12;
            const int gy = RT_R * ty - 1 + drr[k], gx = RT_C * tx - 1 + dcc[k];
            const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const u16* src = ok ? p.x + dsrc[k] + (((long long)b * p.H + gy) * p.W + gx) * 64 : g_zero_page;
            __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(lds + R_X_OFF + kc * RX_CHUNK + pl * RX_PLANE + grp * (16 * ROWB)), 16, 0, 0);
        }
    };

    // ---- 1x1: wave g < 6 owns region pixels 32g .. 32g+31
    const int ridx = wid * 32 + l31;
    const bool rlive = wid < 6 && ridx < RR_PX;
    const int rrr = rlive ? ridx / RR_COLS : 0, rcc = rlive ? ridx - (ridx / RR_COLS) * RR_COLS : 0;
    const int rimg = R_I_OFF + (rrr * RI_RP + rcc) * ROWB;
    const int rsw = (rcc >> 2) & 3;
    int x1a[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) x1a[ks] = R_X_OFF + (wid * 32 + l31) * ROWB + (((ks * 2 + lhi) ^ swz(l31)) * 16);

    // ---- 3x3 fragment addresses
    const int pr = l31 >> 4, pcx = l31 & 15;
    const int pbase = (2 * wm + pr) * RI_RP + pcx;
    int xa[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            xa[ks][kw] = R_I_OFF + (pbase + kw) * ROWB + (((ks * 2 + lhi) ^ (((pcx + kw) >> 2) & 3)) * 16);
    int wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wa[ks] = R_W_OFF + (wn * 32 + l31) * ROWB + (((ks * 2 + lhi) ^ swz(l31)) * 16);

    if ((int)blockIdx.x < p.total) x_dma(blockIdx.x);
    float amax = 0.f;
#ifdef YV3_RES_TL        // debug build only (tools/timeline.py --kernel front): cycle split of one workgroup, written over y[0..]
    unsigned long long tl_t = __builtin_amdgcn_s_memtime(), tl_acc[6] = {0, 0, 0, 0, 0, 0};
#define RTL(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_acc[i] += t_ - tl_t; tl_t = t_; } while (0)
#else
#define RTL(i) do {} while (0)
#endif
    for (int tile = blockIdx.x; tile < p.total; tile += gridDim.x) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int r0 = RT_R * ty, c0 = RT_C * tx;

        // (this tile's x region -- and, first tile, the weights -- were waited for before the previous epilogue's stores
        // were issued, see step 4: the stores themselves drain in the background)
        if (tile == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RTL(0);
        __syncthreads();                                          // everybody's share has landed; every wave is past the previous epilogue
        RTL(1);

        // ---- 2. 1x1 conv (64 -> 32) for the region pixels -> image
        if (wid < 6) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8v xf[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) xf[pl] = *reinterpret_cast<const bf16x8v*>(lds + x1a[ks] + kc * RX_CHUNK + pl * RX_PLANE);
                    acc = mfma_unit<NP>(wf1[kc][ks], xf, acc);
                }
            const int gy = r0 - 1 + rrr, gx = c0 - 1 + rcc;
            const bool inimg = rlive && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = fmaf(acc[4 * g + q], alv1[g][q], bev1[g][q]);
                    v[q] = __builtin_fmaxf(t, 0.1f * t);
                    if constexpr (NP == 2) {
                        amax = __builtin_fmaxf(amax, __builtin_fabsf(v[q]));
                        v[q] = __builtin_amdgcn_fmed3f(v[q], -65504.f, 65504.f);
                    }
                }
                u32x2 qh, ql;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if constexpr (NP == 2) {
                        qh[h] = PlaneOps<2>::pack2_nosat(v[2 * h], v[2 * h + 1]);
                        ql[h] = PlaneOps<2>::pack2_nosat(v[2 * h] - PlaneOps<2>::lo(qh[h]), v[2 * h + 1] - PlaneOps<2>::hi(qh[h]));
                    } else { qh[h] = PlaneOps<1>::pack2(v[2 * h], v[2 * h + 1]); ql[h] = 0u; }
                    if (!inimg) { qh[h] = 0u; ql[h] = 0u; }       // outside the picture: the 3x3 conv's zero padding
                }
                if (rlive) {                                      // channels 8g + 4*lhi .. +3: 8 bytes of slot g
                    *reinterpret_cast<u32x2*>(lds + rimg + ((g ^ rsw) * 16) + lhi * 8) = qh;
                    if constexpr (NP == 2) *reinterpret_cast<u32x2*>(lds + rimg + RI_PLANE + ((g ^ rsw) * 16) + lhi * 8) = ql;
                }
            }
        }
        RTL(2);
        __syncthreads();                                          // image complete; the x region is free again
        RTL(1);
        if (tile + (int)gridDim.x < p.total) x_dma(tile + gridDim.x);          // lands during steps 3-4

        // residual rows of this wave's 32 x 32 output tile (L2-warm: the region DMA just read them), requested before the 3x3
        u32x4 rres[2][NP];
        long long orow[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int t = wm * 32 + ps * 16 + (lane >> 2);
            orow[ps] = (((long long)b * p.H + r0 + (t >> 4)) * p.W + c0 + (t & 15)) * 64 + wn * 32 + (lane & 3) * 8;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) rres[ps][pl] = *reinterpret_cast<const u32x4*>(p.x + pl * p.ps + orow[ps]);
        }

        // ---- 3. 3x3 conv out of LDS
        f32x16 acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8v wf[NP], xf[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    wf[pl] = *reinterpret_cast<const bf16x8v*>(lds + wa[ks] + tap * (NP * 64 * ROWB) + pl * (64 * ROWB));
                    xf[pl] = *reinterpret_cast<const bf16x8v*>(lds + xa[ks][kw] + kh * (RI_RP * ROWB) + pl * RI_PLANE);
                }
                acc2 = mfma_unit<NP>(wf, xf, acc2);
            }
        }
        RTL(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // residual rows + my share of the NEXT tile's x region (long landed)
        RTL(5);
        __syncthreads();                                          // the image is dead: its LDS (+ slack) becomes the transpose tiles
        RTL(1);

        // ---- 4. epilogue: BN + LeakyReLU -> per-wave LDS transpose -> + residual -> hi/lo planes
        float* tl = reinterpret_cast<float*>(lds + R_I_OFF) + wid * (32 * R_EP);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = fmaf(acc2[4 * g + q], alv2[g][q], bev2[g][q]);
                v[q] = __builtin_fmaxf(t, 0.1f * t);
            }
            *reinterpret_cast<f32x4*>(tl + l31 * R_EP + 8 * g + 4 * lhi) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = ps * 16 + (lane >> 2), cg = (lane & 3) * 8;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(tl + r * R_EP + cg);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(tl + r * R_EP + cg + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int h = 0; h < 4; ++h) { v[2 * h] += PlaneOps<NP>::lo(rres[ps][pl][h]); v[2 * h + 1] += PlaneOps<NP>::hi(rres[ps][pl][h]); }
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
                *reinterpret_cast<u32x4*>(p.y + orow[ps]) = qh;
                *reinterpret_cast<u32x4*>(p.y + p.ps + orow[ps]) = ql;
            } else {
                u32x4 qb;
#pragma unroll
                for (int h = 0; h < 4; ++h) qb[h] = PlaneOps<1>::pack2(v[2 * h], v[2 * h + 1]);
                *reinterpret_cast<u32x4*>(p.y + orow[ps]) = qb;
            }
        }
        RTL(4);
    }
#ifdef YV3_RES_TL
    if (blockIdx.x == 17 && lane == 0) {
        float* dbg = reinterpret_cast<float*>(p.y) + wid * 8;
        for (int i = 0; i < 6; ++i) dbg[i] = (float)tl_acc[i];
        dbg[6] = (float)((p.total - 17 + gridDim.x - 1) / gridDim.x);
    }
#endif
    if (p.flags && __any(!(amax <= 65504.f)) && lane == 0) atomicOr(p.flags, 1);
}

}  // namespace

template <int NP>
static int res_block64_launch(const void* x, const void* w1_packed, const float* alpha1, const float* beta1,
                              const void* w2_packed, const float* alpha2, const float* beta2, void* y,
                              int B, int H, int W, int* flags, void* stream) {
    if (!x || !w1_packed || !alpha1 || !beta1 || !w2_packed || !alpha2 || !beta2 || !y || B <= 0 || H <= 0 || W <= 0) return YV3_EINVAL;
    if ((H % RT_R) || (W % RT_C)) return YV3_ESHAPE;                    // whole 8 x 16 tiles only
    Res64Params p;
    p.x = (const u16*)x; p.w1 = (const u16*)w1_packed; p.alpha1 = alpha1; p.beta1 = beta1;
    p.w2 = (const u16*)w2_packed; p.alpha2 = alpha2; p.beta2 = beta2; p.y = (u16*)y;
    p.H = H; p.W = W; p.B = B;
    p.ps = (long long)B * H * W * 64;
    p.tiles_x = W / RT_C; p.tiles_y = H / RT_R;
    const long long total = (long long)B * p.tiles_x * p.tiles_y;
    if (total > 0x7fffffffLL) return YV3_ESHAPE;
    p.total = (int)total;
    p.flags = flags;
    const int ncu = yv3_num_cu();
    const int grid = p.total < ncu ? p.total : ncu;                    // persistent: one workgroup per CU
    hipLaunchKernelGGL(conv_res64_kernel<NP>, dim3(grid), dim3(512), ResGeo<NP>::R_LDS, (hipStream_t)stream, p);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_res_block64(const void* x, const void* w1_packed, const float* alpha1, const float* beta1,
                               const void* w2_packed, const float* alpha2, const float* beta2, void* y,
                               int B, int H, int W, int* flags, void* stream) {
    return res_block64_launch<2>(x, w1_packed, alpha1, beta1, w2_packed, alpha2, beta2, y, B, H, W, flags, stream);
}

extern "C" int yv3_res_block64_bf16(const void* x, const void* w1_packed, const float* alpha1, const float* beta1,
                                    const void* w2_packed, const float* alpha2, const float* beta2, void* y,
                                    int B, int H, int W, int* flags, void* stream) {
    return res_block64_launch<1>(x, w1_packed, alpha1, beta1, w2_packed, alpha2, beta2, y, B, H, W, flags, stream);
}
