// YOLO head decode (reference yololayer.py:31-59,97-105, inference branch) as one fused
// elementwise kernel.  With NHWC logits [B,H,W,3*(5+C)] the reference's final row order
// (y*W + x)*3 + a (yololayer.py:104) IS the input order, so the decode is a pure streaming map:
// one read, one write, no permute.  HBM-bound: 2 * 4 bytes per element.
//
// Op order follows the reference so that results agree to the rounding of exp/sigmoid:
//   xy  : (sigmoid(t) + grid) * stride                yololayer.py:45,58,98
//   wh  : (exp(t) * (anchor / stride)) * stride       yololayer.py:37,59,98
//   conf, cls : sigmoid(t)                            yololayer.py:47-48
#include "yv3_common.h"

namespace {

struct DecodeArgs {
    const float* logits;
    float* out;
    long long out_batch_stride;
    int ld;            // NHWC: floats per pixel; NCHW: unused
    int H, W, attrib;
    float stride;
    float aw[3], ah[3];   // anchor / stride (float32 division, as torch does)
};

// One workgroup iteration = PPB consecutive pixels; thread -> channel ca of the 3*(5+C) per pixel (no
// per-element integer division: anchor / attribute come from two compares, grid x/y once per pixel).
constexpr int PPB = 4;

template <bool NCHW>
__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
    const int b = blockIdx.y;
    const int HW = a.H * a.W;
    const int ch = 3 * a.attrib;
    const float* src = a.logits + (NCHW ? (long long)b * HW * ch : (long long)b * HW * a.ld);
    float* dst = a.out + (long long)b * a.out_batch_stride;
    for (int ca = threadIdx.x; ca < ch; ca += 256) {
        const int anc = (ca >= a.attrib) + (ca >= 2 * a.attrib);
        const int attr = ca - anc * a.attrib;
        const float an = attr == 2 ? a.aw[anc] : a.ah[anc];
        for (int p0 = blockIdx.x * PPB; p0 < HW; p0 += gridDim.x * PPB) {
            float t[PPB];
#pragma unroll
            for (int k = 0; k < PPB; ++k) {
                const int pix = p0 + k < HW ? p0 + k : HW - 1;
                t[k] = NCHW ? src[(long long)ca * HW + pix] : src[(long long)pix * a.ld + ca];
            }
#pragma unroll
            for (int k = 0; k < PPB; ++k) {
                const int pix = p0 + k;
                if (pix >= HW) break;
                dst[(long long)pix * ch + ca] = yv3_decode_value(t[k], attr, an, (float)(pix % a.W), (float)(pix / a.W), a.stride);
            }
        }
    }
}

int run(const float* logits, int ld, const float* anchors, float stride, float* out, long long obs,
        int B, int H, int W, int C, bool nchw, void* stream) {
    if (!logits || !anchors || !out || B <= 0 || H <= 0 || W <= 0 || C < 0 || stride <= 0.f) return YV3_EINVAL;
    DecodeArgs a;
    a.logits = logits; a.out = out; a.out_batch_stride = obs; a.ld = ld;
    a.H = H; a.W = W; a.attrib = 5 + C; a.stride = stride;
    if (!nchw && ld < 3 * a.attrib) return YV3_ESHAPE;
    for (int i = 0; i < 3; ++i) { a.aw[i] = anchors[2 * i] / stride; a.ah[i] = anchors[2 * i + 1] / stride; }
    const int groups = (H * W + PPB - 1) / PPB;
    const dim3 grid((unsigned)(groups < 1024 ? groups : 1024), (unsigned)B);
    if (nchw) hipLaunchKernelGGL(decode_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else      hipLaunchKernelGGL(decode_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    YV3_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int yv3_decode(const float* logits, int ld_logits, const float* anchors_host, float stride,
                          float* out, long long out_batch_stride, int B, int H, int W, int num_class, void* stream) {
    return run(logits, ld_logits, anchors_host, stride, out, out_batch_stride, B, H, W, num_class, false, stream);
}

extern "C" int yv3_decode_nchw(const float* logits_nchw, const float* anchors_host, float stride,
                               float* out, long long out_batch_stride, int B, int H, int W, int num_class, void* stream) {
    return run(logits_nchw, 0, anchors_host, stride, out, out_batch_stride, B, H, W, num_class, true, stream);
}
