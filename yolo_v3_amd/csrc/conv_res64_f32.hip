// The first residual block of Darknet-53 as ONE kernel, EXACT-fp32 mode (YV3_F32):
//     feature.mlist.2 = res_layer(64):  x + conv_bn_relu(32 -> 64, 3x3)(conv_bn_relu(64 -> 32, 1x1)(x))    reference darknet.py:46-53
// The fp32 twin of conv_res64.hip (round 5).  Unfused, the two launches cost 0.23 + 1.09 ms of the exact-fp32 mode's 27 ms step at
// 416x416 bs=64 -- 1.4 GB of fp32 activations through HBM for 0.11 TFLOP -- against 0.72 ms of fp32 MFMA time.  A persistent workgroup per
// CU walks 8x16-pixel output tiles:
//   1. the 10 x 18 x 64-channel region of x the tile needs is DMA-ed into LDS (global_load_lds, issued a tile ahead): 192 rows of 256 B,
//      16-byte slots XOR-swizzled by the row (slot ^ row & 15) so that the MFMA fragment reads (ds_read_b128, 16 lanes = 16 consecutive
//      rows) are conflict-free;
//   2. the 1x1 conv runs on the matrix cores (v_mfma_f32_32x32x2_f32; its weights live in registers) for those 180 pixels, BN + LeakyReLU,
//      written to an LDS-resident fp32 image (zero outside the picture: the 3x3 conv's padding), pitch 20 pixels, rows of 128 B swizzled
//      by the image column (slot ^ (col >> 1) & 7): conflict-free reads for all nine taps;
//   3. the 3x3 conv (M = 128 pixels, N = 64, K = 9 x 32) runs entirely out of LDS -- its weights (72 KB of fp32) are resident for the
//      whole launch -- no global traffic, no barrier inside;
//   4. epilogue: BN + LeakyReLU -> per-wave LDS transpose -> + residual (x rows re-read from L2, requested before step 3) -> 32-byte
//      row segments per lane.
// Same products in the same K order, paired into the MFMA's two k-slots exactly as conv_igemm_f32_kernel pairs them (k = 8 kk + t and
// 8 kk + 4 + t), same epilogue operations: BIT-IDENTICAL to yv3_conv2d (1x1) followed by yv3_conv2d (3x3 + residual) in YV3_F32
// (tests/test_gpu_kernels.py::test_fused_res64_f32_equals_two_launches_bitwise).  HBM traffic: x once (+ 41 % halo) + y once.
#include "yv3_common.h"

namespace {

constexpr int QT_R = 8, QT_C = 16;                        // output tile (rows x cols)
constexpr int QR_COLS = QT_C + 2, QR_PX = (QT_R + 2) * QR_COLS;   // region 10 x 18 = 180 pixels
constexpr int QR_ROWS = 192;                              // padded to 6 MFMA row blocks
constexpr int QX_ROWB = 64 * 4;                           // bytes per x-region row (64 channels)
constexpr int QX_BYTES = QR_ROWS * QX_ROWB;               // 49 152
constexpr int QI_RP = 20;                                 // image pitch (pixels per row; even: pixel parity == column parity)
constexpr int QI_ROWB = 32 * 4;                           // bytes per image pixel (32 channels)
constexpr int QI_BYTES = (QT_R + 2) * QI_RP * QI_ROWB;    // 25 600
constexpr int Q_EP = 36;                                  // floats per row of a wave's epilogue transpose tile
constexpr int QE_BYTES = 8 * 32 * Q_EP * 4;               // 36 864 (re-uses the image's space + slack)
constexpr int QW_BYTES = 9 * 64 * QI_ROWB;                // 73 728: [tap][64 channel rows][32 k]
constexpr int Q_X_OFF = 0, Q_I_OFF = QX_BYTES, Q_W_OFF = Q_I_OFF + QE_BYTES, Q_LDS = Q_W_OFF + QW_BYTES;   // 159 744
static_assert(QI_BYTES <= QE_BYTES && Q_LDS <= 160 * 1024, "LDS budget");

#define QGPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define QLPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __attribute__((aligned(64))) float g_zero_f32_res[16];      // zero-initialised: source of out-of-picture region rows

struct Res64F32Params {
    const float* x;          // [B,H,W,64]
    const float* w1; const float* alpha1; const float* beta1;      // 1x1 64 -> 32: [32][64]
    const float* w2; const float* alpha2; const float* beta2;      // 3x3 32 -> 64: [64][3][3][32]
    float* y;                // [B,H,W,64]
    int H, W, B, tiles_x, tiles_y, total;
};

__global__ __launch_bounds__(512) void conv_res64_f32_kernel(const Res64F32Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;                // 3x3 wave tile: 32 pixels (tile rows 2wm, 2wm+1) x 32 channels

    // ---- 3x3 weights: resident in LDS for the launch.  72 wave instructions of 8 rows x 128 B; lane -> (row, physical slot)
    for (int pc = wid; pc < QW_BYTES / 1024; pc += 8) {
        const int r = pc * 8 + (lane >> 3);               // tap * 64 + channel row
        const int tap = r >> 6, n = r & 63;
        const int ls = (lane & 7) ^ ((n >> 1) & 7);       // logical 16-byte slot this lane carries
        __builtin_amdgcn_global_load_lds(QGPTR(p.w2 + (long long)n * 288 + tap * 32 + ls * 4), QLPTR(lds + Q_W_OFF + pc * 1024), 16, 0, 0);
    }
    // ---- 1x1 weights: this lane's operands, resident in registers: channel l31, k = 32 chunk + 8 kk + 4 lhi + t
    f32x4 w1r[2][4];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w1r[kc][kk] = *reinterpret_cast<const f32x4*>(p.w1 + l31 * 64 + kc * 32 + kk * 8 + lhi * 4);
    const float al1 = p.alpha1[l31], be1 = p.beta1[l31];
    const float al2 = p.alpha2[wn * 32 + l31], be2 = p.beta2[wn * 32 + l31];

    // ---- x-region DMA: 48 wave instructions per tile (4 rows of 256 B each); this wave issues i = wid + 8k.  Per lane: region pixel of
    // its row, logical slot (the swizzle is applied on the source side), all tile-independent.
    constexpr int DK = 6;
    int drr[DK], dcc[DK], dls[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) {
        const int row = (wid + 8 * k) * 4 + (lane >> 4);
        drr[k] = row < QR_PX ? row / QR_COLS : -100;      // rows 180..191: never inside the picture -> zero page
        dcc[k] = row - (row / QR_COLS) * QR_COLS;
        dls[k] = ((lane & 15) ^ (row & 15)) * 4;          // floats
    }
    auto x_dma = [&](int tile) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            const int gy = QT_R * ty - 1 + drr[k], gx = QT_C * tx - 1 + dcc[k];
            const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const float* src = ok ? p.x + (((long long)b * p.H + gy) * p.W + gx) * 64 + dls[k] : g_zero_f32_res;
            __builtin_amdgcn_global_load_lds(QGPTR(src), QLPTR(lds + Q_X_OFF + (wid + 8 * k) * 1024), 16, 0, 0);
        }
    };

    // ---- 1x1: wave g < 6 owns region pixels 32g .. 32g+31 (A operand rows); its D tile: channel l31, pixels (e&3) + 8 (e>>2) + 4 lhi
    const int x1a = Q_X_OFF + (wid * 32 + l31) * QX_ROWB;                  // + ((slot ^ (row & 15)) * 16), row & 15 == l31 & 15
    const int x1sw = l31 & 15;
    int himg[16], hpos[16];                                              // image byte address (-1: no pixel) and (row << 8 | col) of the D pixels
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ridx = wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        const bool live = wid < 6 && ridx < QR_PX;
        const int rr = live ? ridx / QR_COLS : 0, cc = live ? ridx - (ridx / QR_COLS) * QR_COLS : 0;
        himg[e] = live ? Q_I_OFF + (rr * QI_RP + cc) * QI_ROWB + (((l31 >> 2) ^ ((cc >> 1) & 7)) * 16) + (l31 & 3) * 4 : -1;
        hpos[e] = (rr << 8) | cc;
    }

    // ---- 3x3 fragment addresses: pixel side per column tap (the swizzle follows the image column), weight side
    const int pr = l31 >> 4, pcx = l31 & 15;
    const int pbase = (2 * wm + pr) * QI_RP + pcx;
    int xa[3], xsw[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) { xa[kw] = Q_I_OFF + (pbase + kw) * QI_ROWB; xsw[kw] = ((pcx + kw) >> 1) & 7; }
    const int wa = Q_W_OFF + (wn * 32 + l31) * QI_ROWB, wsw = (l31 >> 1) & 7;          // ((wn * 32 + l31) >> 1) & 7 == (l31 >> 1) & 7

    if ((int)blockIdx.x < p.total) x_dma(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.total; tile += gridDim.x) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int r0 = QT_R * ty, c0 = QT_C * tx;

        // (this tile's x region -- and, first tile, the weights -- were waited for before the previous epilogue's stores were issued,
        // see step 4: the stores themselves drain in the background)
        if (tile == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // everybody's share has landed; every wave is past the previous epilogue

        // ---- 2. 1x1 conv (64 -> 32) for the region pixels -> image
        if (wid < 6) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 af = *reinterpret_cast<const f32x4*>(lds + x1a + (((kc * 8 + kk * 2 + lhi) ^ x1sw) * 16));
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], w1r[kc][kk][t], acc, 0, 0, 0);
                }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (himg[e] >= 0) {
                    const int gy = r0 - 1 + (hpos[e] >> 8), gx = c0 - 1 + (hpos[e] & 255);
                    float v = fmaf(acc[e], al1, be1);
                    v = v > 0.f ? v : 0.1f * v;                   // (conv_igemm_f32.hip's form of LeakyReLU(0.1))
                    if (!((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)) v = 0.f;      // the 3x3 conv's zero padding
                    *reinterpret_cast<float*>(lds + himg[e]) = v;
                }
            }
        }
        __syncthreads();                                          // image complete; the x region is free again
        if (tile + (int)gridDim.x < p.total) x_dma(tile + gridDim.x);          // lands during steps 3-4

        // residual rows of this wave's 32 x 32 output tile (L2-warm: the region DMA just read them), requested before the 3x3
        f32x4 rres[2][2];
        long long orow[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int t = wm * 32 + ps * 16 + (lane >> 2);
            orow[ps] = (((long long)b * p.H + r0 + (t >> 4)) * p.W + c0 + (t & 15)) * 64 + wn * 32 + (lane & 3) * 8;
            rres[ps][0] = *reinterpret_cast<const f32x4*>(p.x + orow[ps]);
            rres[ps][1] = *reinterpret_cast<const f32x4*>(p.x + orow[ps] + 4);
        }

        // ---- 3. 3x3 conv out of LDS: 9 taps x 4 groups of 8 k, K order (kh, kw, c)
        f32x16 acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(lds + xa[kw] + kh * (QI_RP * QI_ROWB) + (((kk * 2 + lhi) ^ xsw[kw]) * 16));
                const f32x4 bf = *reinterpret_cast<const f32x4*>(lds + wa + tap * (64 * QI_ROWB) + (((kk * 2 + lhi) ^ wsw) * 16));
#pragma unroll
                for (int t = 0; t < 4; ++t) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bf[t], acc2, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // residual rows + my share of the NEXT tile's x region (long landed)
        __syncthreads();                                          // the image is dead: its LDS (+ slack) becomes the transpose tiles

        // ---- 4. epilogue: BN + LeakyReLU -> per-wave LDS transpose -> + residual -> 32-byte row segments
        float* tl = reinterpret_cast<float*>(lds + Q_I_OFF) + wid * (32 * Q_EP);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = fmaf(acc2[e], al2, be2);
            v = v > 0.f ? v : 0.1f * v;
            tl[((e & 3) + 8 * (e >> 2) + 4 * lhi) * Q_EP + l31] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = ps * 16 + (lane >> 2), cg = (lane & 3) * 8;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(tl + r * Q_EP + cg);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(tl + r * Q_EP + cg + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v0[q] += rres[ps][0][q]; v1[q] += rres[ps][1][q]; }
            *reinterpret_cast<f32x4*>(p.y + orow[ps]) = v0;
            *reinterpret_cast<f32x4*>(p.y + orow[ps] + 4) = v1;
        }
    }
}

}  // namespace

extern "C" int yv3_res_block64_f32(const float* x, const float* w1_packed, const float* alpha1, const float* beta1,
                                   const float* w2_packed, const float* alpha2, const float* beta2, float* y,
                                   int B, int H, int W, void* stream) {
    if (!x || !w1_packed || !alpha1 || !beta1 || !w2_packed || !alpha2 || !beta2 || !y || B <= 0 || H <= 0 || W <= 0) return YV3_EINVAL;
    if ((H % QT_R) || (W % QT_C)) return YV3_ESHAPE;                    // whole 8 x 16 tiles only
    Res64F32Params p;
    p.x = x; p.w1 = w1_packed; p.alpha1 = alpha1; p.beta1 = beta1;
    p.w2 = w2_packed; p.alpha2 = alpha2; p.beta2 = beta2; p.y = y;
    p.H = H; p.W = W; p.B = B;
    p.tiles_x = W / QT_C; p.tiles_y = H / QT_R;
    const long long total = (long long)B * p.tiles_x * p.tiles_y;
    if (total > 0x7fffffffLL) return YV3_ESHAPE;
    p.total = (int)total;
    const int ncu = yv3_num_cu();
    const int grid = p.total < ncu ? p.total : ncu;                    // persistent: one workgroup per CU
    hipLaunchKernelGGL(conv_res64_f32_kernel, dim3(grid), dim3(512), Q_LDS, (hipStream_t)stream, p);
    YV3_CHECK_LAUNCH();
    return 0;
}
