// Winograd F(2x2,3x3) input transform for the fp16 hi+lo plane mode (yv3_conv_desc.w_wino, include/yv3.h).
//
// A 3x3 / stride-1 / pad-1 convolution (reference darknet.py:43-44 for the conv_bn_relu blocks, :52-53 inside res_layer)
// over 2x2 output tiles:  Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A  with
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// This kernel produces V = B^T d B for every tile and channel: it reads the 4x4 input patch (rows 2ty-1 .. 2ty+2, zero
// outside the picture) from the two fp16 planes, rebuilds the fp32 value hi + lo, transforms in fp32 (adds / subtracts only),
// scales by 1/4 (exact; keeps |V| inside the fp16 range whenever the input is: |V| <= 4 max|d|) and splits the result into
// hi / lo planes again.  Output layout: [2 planes][16 positions][T tiles][C] fp16, T = B * ceil(H/2) * ceil(W/2) -- for a
// fixed position the tiles x channels matrix is contiguous, i.e. exactly the A operand of a 1x1-convolution-style GEMM,
// which conv_planes.hip's WINO main loop streams with the same DMA pieces as any other layer.
// HBM-bound: reads 4 B and writes 16 B per input element.
#include "conv_planes_common.h"

namespace {

// one thread = one tile x 8 channels (16-byte loads / stores; a wave covers 64 * 8 consecutive channels-of-tiles)
__global__ __launch_bounds__(256) void wino_input_kernel(const u16* __restrict__ x, long long xs, u16* __restrict__ v, long long vs,
                                                         int H, int W, int C, int th, int tw, long long T) {
    const int cg = C >> 3;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * cg) return;
    const long long t = i / cg;
    const int c = (int)(i - t * cg) * 8;
    const int b = (int)(t / (th * tw));
    const int rem = (int)(t - (long long)b * th * tw);
    const int ty = rem / tw, tx = rem - ty * tw;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    float d[4][4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = y0 + r, xx = x0 + q;
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            u32x4 h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
            if (ok) {
                const long long o = (((long long)b * H + yy) * W + xx) * C + c;
                h = *reinterpret_cast<const u32x4*>(x + o);
                l = *reinterpret_cast<const u32x4*>(x + xs + o);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d[r][q][2 * e] = PlaneOps<2>::lo(h[e]) + PlaneOps<2>::lo(l[e]);
                d[r][q][2 * e + 1] = PlaneOps<2>::hi(h[e]) + PlaneOps<2>::hi(l[e]);
            }
        }
    // rows: t = B^T d
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d0 = d[0][q][e], d1 = d[1][q][e], d2 = d[2][q][e], d3 = d[3][q][e];
            d[0][q][e] = d0 - d2; d[1][q][e] = d1 + d2; d[2][q][e] = d2 - d1; d[3][q][e] = d1 - d3;
        }
    // columns: V = t B, x 1/4, split, store
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float o[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t0 = d[r][0][e], t1 = d[r][1][e], t2 = d[r][2][e], t3 = d[r][3][e];
            o[0][e] = 0.25f * (t0 - t2); o[1][e] = 0.25f * (t1 + t2); o[2][e] = 0.25f * (t2 - t1); o[3][e] = 0.25f * (t1 - t3);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 qh, ql;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qh[e] = PlaneOps<2>::pack2(o[q][2 * e], o[q][2 * e + 1]);
                ql[e] = PlaneOps<2>::pack2_nosat(o[q][2 * e] - PlaneOps<2>::lo(qh[e]), o[q][2 * e + 1] - PlaneOps<2>::hi(qh[e]));
            }
            const long long o_ = ((long long)(r * 4 + q) * T + t) * C + c;
            *reinterpret_cast<u32x4*>(v + o_) = qh;
            *reinterpret_cast<u32x4*>(v + vs + o_) = ql;
        }
    }
}

// fp32 mode (YV3_F32): x NHWC fp32 -> V [16][T][C] fp32, no scaling (fp32 has the range); one thread = one tile x 4 channels
__global__ __launch_bounds__(256) void wino_input_f32_kernel(const float* __restrict__ x, float* __restrict__ v,
                                                             int H, int W, int C, int th, int tw, long long T) {
    const int cg = C >> 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * cg) return;
    const long long t = i / cg;
    const int c = (int)(i - t * cg) * 4;
    const int b = (int)(t / (th * tw));
    const int rem = (int)(t - (long long)b * th * tw);
    const int ty = rem / tw, tx = rem - ty * tw;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    f32x4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = y0 + r, xx = x0 + q;
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            d[r][q] = ok ? *reinterpret_cast<const f32x4*>(x + (((long long)b * H + yy) * W + xx) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 d0 = d[0][q], d1 = d[1][q], d2 = d[2][q], d3 = d[3][q];
        d[0][q] = d0 - d2; d[1][q] = d1 + d2; d[2][q] = d2 - d1; d[3][q] = d1 - d3;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const f32x4 t0 = d[r][0], t1 = d[r][1], t2 = d[r][2], t3 = d[r][3];
        const f32x4 o[4] = {t0 - t2, t1 + t2, t2 - t1, t1 - t3};
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(v + ((long long)(r * 4 + q) * T + t) * C + c) = o[q];
    }
}

}  // namespace

extern "C" size_t yv3_wino_workspace_bytes(int B, int H, int W, int cin) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0) return 0;
    // transformed input + the hand-over area of the even (stream-K) schedule (parts + flags; must start zero-filled)
    return (((size_t)2 * 16 * B * ((H + 1) / 2) * ((W + 1) / 2) * cin * sizeof(u16) + 255) & ~(size_t)255) + yv3_wino_sk_bytes();
}

// V = B^T d B of the [2][B,H,W,C] fp16-plane tensor x (plane stride xs elements) -> v = [2][16][T][C]
int yv3_wino_input_transform(const u16* x, long long xs, u16* v, int B, int H, int W, int C, hipStream_t s) {
    if ((C & 7) || B <= 0) return YV3_ESHAPE;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = (long long)B * th * tw;
    const long long n = T * (C >> 3);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, xs, v, 16 * T * C, H, W, C, th, tw, T);
    YV3_CHECK_LAUNCH();
    return 0;
}

// fp32 form: x NHWC fp32 [B,H,W,C] -> v = [16][T][C] fp32
int yv3_wino_input_transform_f32(const float* x, float* v, int B, int H, int W, int C, hipStream_t s) {
    if ((C & 3) || B <= 0) return YV3_ESHAPE;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = (long long)B * th * tw;
    const long long n = T * (C >> 2);
    hipLaunchKernelGGL(wino_input_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, v, H, W, C, th, tw, T);
    YV3_CHECK_LAUNCH();
    return 0;
}
