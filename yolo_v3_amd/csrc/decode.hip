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

__device__ inline float sigmoidf_(float t) { return 1.f / (1.f + expf(-t)); }

template <bool NCHW>
__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
    const int b = blockIdx.y;
    const int HW = a.H * a.W;
    const int ch = 3 * a.attrib;
    const long long per_batch = (long long)HW * ch;
    const float* src = a.logits + (NCHW ? (long long)b * per_batch : (long long)b * HW * a.ld);
    float* dst = a.out + (long long)b * a.out_batch_stride;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < per_batch; r += (long long)gridDim.x * 256) {
        const int pix = (int)(r / ch);
        const int ca = (int)(r - (long long)pix * ch);
        const int anc = ca / a.attrib;
        const int attr = ca - anc * a.attrib;
        const float t = NCHW ? src[(long long)ca * HW + pix] : src[(long long)pix * a.ld + ca];
        float v;
        if (attr >= 4) {
            v = sigmoidf_(t);
        } else if (attr == 0) {
            v = (sigmoidf_(t) + (float)(pix % a.W)) * a.stride;
        } else if (attr == 1) {
            v = (sigmoidf_(t) + (float)(pix / a.W)) * a.stride;
        } else if (attr == 2) {
            v = (expf(t) * a.aw[anc]) * a.stride;
        } else {
            v = (expf(t) * a.ah[anc]) * a.stride;
        }
        dst[r] = v;
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
    const long long per_batch = (long long)H * W * 3 * a.attrib;
    const int bx = (int)((per_batch + 255) / 256 < 2048 ? (per_batch + 255) / 256 : 2048);
    const dim3 grid((unsigned)bx, (unsigned)B);
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
