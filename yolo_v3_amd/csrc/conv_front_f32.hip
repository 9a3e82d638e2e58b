// The first two layers of Darknet-53 as ONE kernel, EXACT-fp32 mode (YV3_F32) -- the fp32 twin of conv_front.hip (round 5):
//     feature.mlist.0  conv_bn_relu(3 -> 32, 3x3, s1)   reference darknet.py:76, :43-44
//     feature.mlist.1  conv_bn_relu(32 -> 64, 3x3, s2)  reference darknet.py:68-70 (make_res_stack's down-sampling conv)
// Unfused, the first layer writes a [B,H,W,32] fp32 activation (1.42 GB at 416x416 bs=64) that the second immediately re-reads:
// 0.68 + 1.04 ms of the mode's 26.5 ms step.  Here it never leaves the CU: a persistent workgroup walks 8x16-pixel tiles of the
// SECOND layer's output and for each tile
//   1. stages the 19 x 35 x 3 input patch (NCHW fp32, zero halo) in LDS                  (prefetched one tile ahead in registers)
//   2. computes the 17 x 33 first-layer pixels the tile needs on the vector ALUs -- conv0.hip's conv0_kernel<0> chain, fma for fma:
//      acc = fma(x, w, acc) over (c, kh, kw), then BN + LeakyReLU.  Wave = (pixel block, channel octet): the octet's weights are
//      wave-uniform (scalar loads), a lane owns one pixel x 8 channels (v_pk_fma_f32), 9 passes of 64 pixels per SIMD.  Written to
//      an LDS-resident fp32 image, zero where the pixel lies outside the picture (the second conv's padding), stored by column
//      parity [parity][17 rows][18] so that the stride-2 taps read CONSECUTIVE 128-byte rows, XOR-swizzled by the half-column
//      (slot ^ (colh >> 1) & 7): conflict-free ds_read_b128 for all nine taps;
//   3. runs the second conv (M = 128 pixels, N = 64, K = 9 taps x 32; v_mfma_f32_32x32x2_f32) entirely out of LDS: its weights
//      (72 KB of fp32) are DMA-ed once per workgroup and stay resident -- no global traffic, no barrier inside;
//   4. BN + LeakyReLU -> per-wave LDS transpose -> 32-byte row segments per lane.
// Same operations in the same order as yv3_conv0(YV3_F32) followed by yv3_conv2d(YV3_F32) (k pairs 8 kk + t / 8 kk + 4 + t per
// MFMA, as conv_igemm_f32_kernel): BIT-IDENTICAL to the two launches (tests/test_gpu_kernels.py::
// test_fused_front_f32_equals_two_launches_bitwise).  HBM traffic: 12 B in + 256 B out per second-layer pixel.
// LDS: 78 336 (image, re-used by the epilogue transposes) + 73 728 (weights) + 8 208 (patch) = 160 272 B: one workgroup per CU.
#include "yv3_common.h"

namespace {

constexpr int GT_R = 8, GT_C = 16;                        // output tile of the second conv (rows x cols)
constexpr int GR_COLS = 2 * GT_C + 1;                     // first-layer region: 17 rows x 33 cols
constexpr int GR_PX = (2 * GT_R + 1) * GR_COLS;           // 561
constexpr int GP_ROWS = 2 * GT_R + 3, GP_COLS = 2 * GT_C + 3;   // input patch 19 x 35
constexpr int GP_PITCH = 36, GP_CH = GP_ROWS * GP_PITCH;  // floats
constexpr int GA_RP = 18, GA_PB = (2 * GT_R + 1) * GA_RP; // image row pitch (pixels), parity block (306 pixels; even)
constexpr int G_ROWB = 32 * 4;                            // bytes per image pixel / per weight row (32 fp32)
constexpr int GA_BYTES = 2 * GA_PB * G_ROWB;              // 78 336
constexpr int GW_BYTES = 9 * 64 * G_ROWB;                 // 73 728: [tap][64 channel rows][32 k]
constexpr int G_EP = 36;                                  // floats per row of a wave's epilogue transpose tile
constexpr int G_A_OFF = 0, G_W_OFF = GA_BYTES, G_P_OFF = G_W_OFF + GW_BYTES, G_LDS = G_P_OFF + 3 * GP_CH * 4;     // 160 272
static_assert(8 * 32 * G_EP * 4 <= GA_BYTES && G_LDS <= 160 * 1024, "LDS budget");

#define GGPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define GLPTR(p) ((__attribute__((address_space(3))) void*)(p))

typedef float gf32x2 __attribute__((ext_vector_type(2)));

struct FrontF32Params {
    const float* x;          // [B,3,H,W] fp32
    const float* w0;         // first layer weights [27][32] fp32 (tap-major: ((c*3 + kh)*3 + kw)*32 + n)
    const float* alpha0; const float* beta0;
    const float* w1;         // second layer [64][3][3][32] fp32
    const float* alpha1; const float* beta1;
    float* y;                // [B,H/2,W/2,64]
    int H, W, B, tiles_x, tiles_y, total;
};

// (w0 is a __restrict__ kernel argument of its own: only then are the wave-uniform weight reads provably unclobbered -> scalar loads)
__global__ __launch_bounds__(512) void conv_front_f32_kernel(const FrontF32Params p, const float* __restrict__ w0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* const patch = reinterpret_cast<float*>(lds + G_P_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);                // 8 waves: two per SIMD
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;                // second conv: wave tile 32 pixels (tile rows 2wm, 2wm+1) x 32 channels
    const int pb = wid >> 2, cq = wid & 3;                // first layer: pixel block (of 64) within a pass of 128, channel octet
    const int Ho = p.H >> 1, Wo = p.W >> 1;

    // ---- second-layer weights: resident in LDS for the launch.  72 wave instructions of 8 rows x 128 B; lane -> (row, physical slot)
    for (int pc = wid; pc < GW_BYTES / 1024; pc += 8) {
        const int r = pc * 8 + (lane >> 3);               // tap * 64 + channel row
        const int tap = r >> 6, n = r & 63;
        const int ls = (lane & 7) ^ ((n >> 1) & 7);       // logical 16-byte slot this lane carries
        __builtin_amdgcn_global_load_lds(GGPTR(p.w1 + (long long)n * 288 + tap * 32 + ls * 4), GLPTR(lds + G_W_OFF + pc * 1024), 16, 0, 0);
    }
    const float* const w0q = w0 + cq * 8;               // this wave's channel octet (wave-uniform: scalar loads)
    float al0[8], be0[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { al0[j] = p.alpha0[cq * 8 + j]; be0[j] = p.beta0[cq * 8 + j]; }
    const float al1 = p.alpha1[wn * 32 + l31], be1 = p.beta1[wn * 32 + l31];

    // ---- fragment addresses of the second conv (constant over tiles).  Pixel side: lane -> (tile row, tile col) of its output pixel;
    // image pixel of tap (kh,kw): parity = kw&1, row 2r+kh, half-column c + (kw>>1); the swizzle key follows the half-column
    const int pr = l31 >> 4, pcx = l31 & 15;
    int xa[2], xsw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { xa[q] = G_A_OFF + ((2 * (2 * wm + pr)) * GA_RP + pcx + q) * G_ROWB; xsw[q] = ((pcx + q) >> 1) & 7; }
    const int wa = G_W_OFF + (wn * 32 + l31) * G_ROWB, wsw = (l31 >> 1) & 7;

    // ---- patch elements of this thread (i = tid + 512*k of 3 x 19 x 35): LDS index, offset in the picture, position -- tile-independent
    constexpr int PN = 3 * GP_ROWS * GP_COLS;             // 1995
    constexpr int PK = (PN + 511) / 512;                  // 4
    int plds[PK], prr[PK], pcc[PK];
    long long pgo[PK];
#pragma unroll
    for (int k = 0; k < PK; ++k) {
        const int i = tid + 512 * k;
        const int c = i / (GP_ROWS * GP_COLS);
        const int r2 = i - c * (GP_ROWS * GP_COLS);
        prr[k] = r2 / GP_COLS; pcc[k] = r2 - prr[k] * GP_COLS;
        plds[k] = i < PN ? c * GP_CH + prr[k] * GP_PITCH + pcc[k] : -1;
        pgo[k] = ((long long)c * p.H + prr[k] - 2) * p.W + pcc[k] - 2;
    }
    float pre[PK];
    auto patch_fetch = [&](int tile) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const float* xb = p.x + (size_t)b * 3 * p.H * p.W + (long long)(2 * GT_R * ty) * p.W + 2 * GT_C * tx;
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            const int gy = 2 * GT_R * ty - 2 + prr[k], gx = 2 * GT_C * tx - 2 + pcc[k];
            float v = 0.f;
            if (plds[k] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = xb[pgo[k]];
            pre[k] = v;
        }
    };
    if ((int)blockIdx.x < p.total) patch_fetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < p.total; tile += gridDim.x) {
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int rem = tile - b * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int r0 = GT_R * ty, c0 = GT_C * tx;

        // ---- 1. patch -> LDS.  Every wave is past the previous tile's epilogue here.
#pragma unroll
        for (int k = 0; k < PK; ++k)
            if (plds[k] >= 0) patch[plds[k]] = pre[k];
        if (tile == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my share of the weight DMA has landed
        __syncthreads();
        if (tile + (int)gridDim.x < p.total) patch_fetch(tile + gridDim.x);       // lands during steps 2-4

        // ---- 2. first layer for the 561 region pixels: passes of 128 pixels (this wave: 64 of them, 8 channels each)
#pragma unroll 1
        for (int ps = 0; ps < (GR_PX + 127) / 128; ++ps) {
            const int base = ps * 128 + pb * 64;
            if (base >= GR_PX) break;                                           // (wave-uniform)
            const int idx = base + lane;
            const bool live = idx < GR_PX;
            const int ii = live ? idx : 0;
            const int row = ii / GR_COLS, col = ii - row * GR_COLS;
            const float* p0 = patch + row * GP_PITCH + col;
            gf32x2 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = gf32x2{0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float v = p0[c * GP_CH + kh * GP_PITCH + kw];
                        const float* wr = w0q + ((c * 3 + kh) * 3 + kw) * 32;       // wave-uniform -> s_load
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[j] = __builtin_elementwise_fma(gf32x2{v, v}, gf32x2{wr[2 * j], wr[2 * j + 1]}, acc[j]);
                    }
            const int gy = 2 * r0 - 1 + row, gx = 2 * c0 - 1 + col;
            const bool inimg = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            f32x4 o[2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = fmaf(acc[j >> 1][j & 1], al0[j], be0[j]);
                v = v > 0.f ? v : 0.1f * v;                                      // (conv0_kernel's form of LeakyReLU(0.1))
                o[j >> 2][j & 3] = inimg ? v : 0.f;                             // outside the picture: the second conv's zero padding
            }
            if (live) {
                const int colh = col >> 1, key = (colh >> 1) & 7;
                unsigned char* d = lds + G_A_OFF + ((col & 1) * GA_PB + row * GA_RP + colh) * G_ROWB;
                *reinterpret_cast<f32x4*>(d + (((2 * cq) ^ key) * 16)) = o[0];
                *reinterpret_cast<f32x4*>(d + (((2 * cq + 1) ^ key) * 16)) = o[1];
            }
        }
        __syncthreads();

        // ---- 3. second conv out of LDS: 9 taps x 4 groups of 8 k, K order (kh, kw, c)
        f32x16 acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            const int aoff = ((kw & 1) * GA_PB + kh * GA_RP) * G_ROWB;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(lds + xa[kw >> 1] + aoff + (((kk * 2 + lhi) ^ xsw[kw >> 1]) * 16));
                const f32x4 bf = *reinterpret_cast<const f32x4*>(lds + wa + tap * (64 * G_ROWB) + (((kk * 2 + lhi) ^ wsw) * 16));
#pragma unroll
                for (int t = 0; t < 4; ++t) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bf[t], acc2, 0, 0, 0);
            }
        }
        __syncthreads();                                                              // the image is dead: its LDS becomes the transpose tiles

        // ---- 4. epilogue: BN + LeakyReLU -> per-wave LDS transpose -> 32-byte row segments
        float* tl = reinterpret_cast<float*>(lds + G_A_OFF) + wid * (32 * G_EP);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = fmaf(acc2[e], al1, be1);
            v = v > 0.f ? v : 0.1f * v;
            tl[((e & 3) + 8 * (e >> 2) + 4 * lhi) * G_EP + l31] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int r = s * 16 + (lane >> 2), cg = (lane & 3) * 8;
            const int t = wm * 32 + r;                                                // pixel of the 8 x 16 tile
            const long long m = ((long long)b * Ho + r0 + (t >> 4)) * Wo + c0 + (t & 15);
            float* yo = p.y + m * 64 + wn * 32 + cg;
            *reinterpret_cast<f32x4*>(yo) = *reinterpret_cast<const f32x4*>(tl + r * G_EP + cg);
            *reinterpret_cast<f32x4*>(yo + 4) = *reinterpret_cast<const f32x4*>(tl + r * G_EP + cg + 4);
        }
    }
}

}  // namespace

extern "C" int yv3_conv_front_f32(const float* x_nchw, const float* w0_tap_major, const float* alpha0, const float* beta0,
                                  const float* w1_packed, const float* alpha1, const float* beta1, float* y,
                                  int B, int H, int W, void* stream) {
    if (!x_nchw || !w0_tap_major || !alpha0 || !beta0 || !w1_packed || !alpha1 || !beta1 || !y || B <= 0 || H <= 0 || W <= 0)
        return YV3_EINVAL;
    if ((H % (2 * GT_R)) || (W % (2 * GT_C))) return YV3_ESHAPE;          // whole 8 x 16 output tiles only (network inputs are multiples of 32)
    FrontF32Params p;
    p.x = x_nchw; p.w0 = w0_tap_major; p.alpha0 = alpha0; p.beta0 = beta0;
    p.w1 = w1_packed; p.alpha1 = alpha1; p.beta1 = beta1; p.y = y;
    p.H = H; p.W = W; p.B = B;
    p.tiles_x = (W / 2) / GT_C; p.tiles_y = (H / 2) / GT_R;
    const long long total = (long long)B * p.tiles_x * p.tiles_y;
    if (total > 0x7fffffffLL) return YV3_ESHAPE;
    p.total = (int)total;
    const int ncu = yv3_num_cu();
    const int grid = p.total < ncu ? p.total : ncu;                      // persistent: one workgroup per CU
    hipLaunchKernelGGL(conv_front_f32_kernel, dim3(grid), dim3(512), G_LDS, (hipStream_t)stream, p, w0_tap_major);
    YV3_CHECK_LAUNCH();
    return 0;
}
