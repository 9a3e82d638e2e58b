// The steps immediately before and after the hot path ("next" rows 1-2):
//   letterbox_kernel      reference utils.py:34-72 (letterbox_transforms / letterbox_image / load_image):
//                         uint8 HWC RGB -> bicubic resize keeping aspect -> centred on a 128-grey canvas ->
//                         /255 -> fp32 CHW, written straight into the network's NCHW input batch
//   resize_linear_kernel  reference utils.py:68-71 (load_image mode='resize': cv2.resize(img, dim), INTER_LINEAR)
//   correct_boxes_kernel  reference boundingbox.py:95-149 (letterbox_reverse / rescale_bbox /
//                         correct_yolo_boxes): x1y1x2y2 in network pixels -> clipped xywh in the ORIGINAL image
#include "yv3_common.h"

namespace {

// cv2.resize for CV_8U, fixed-point path (OpenCV modules/imgproc/src/resize.cpp; cv2 itself is absent from this
// environment, so parity with it is UNPINNED -- the CPU oracle restates exactly this integer algorithm and the
// kernels match it bit for bit):
//   coordinates   fx = (float)((dx + 0.5) * scale - 0.5), scale = 1 / ((double)dst / src); sx = floor(fx); fx -= sx
//   INTER_CUBIC   interpolateCubic (A = -0.75, float32, source operation order) -> saturate_cast<short>(c * 2048);
//                 HResizeCubic: int32 sum of 4 taps (replicated border); VResizeCubic + FixedPtCast<int,uchar,22>:
//                 saturate_cast<uchar>((sum + (1 << 21)) >> 22)
//   INTER_LINEAR  (1 - fx, fx) * 2048 as shorts, edge clamps; VResizeLinear<uchar,int,short>:
//                 uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2); exact 2x2 down-scale ->
//                 INTER_AREA (a + b + c + d + 2) >> 2
// (this translation unit is compiled with -ffp-contract=off: the float32 coefficient arithmetic must not fuse)
__device__ inline void cv_coord(int d, double scale, int& s, float& f) {
    const float fx = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(fx);
    f = fx - (float)s;
}
__device__ inline int cv_short(float c) {                      // saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE)
    return min(max((int)rintf(c * 2048.f), -32768), 32767);
}
__device__ inline void cv_cubic_coeffs(float x, int c[4]) {
    const float A = -0.75f;
    float w[4];
    w[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    w[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = cv_short(w[k]);
}

__global__ __launch_bounds__(256) void letterbox_kernel(const unsigned char* __restrict__ img, int H, int W,
                                                       float* __restrict__ out, int OH, int OW,
                                                       int box_w, int box_h, int box_x, int box_y,
                                                       double scale_x, double scale_y) {
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= OH * OW) return;
    const int oy = px / OW, ox = px - oy * OW;
    int rgb[3] = {128, 128, 128};
    const int bx = ox - box_x, by = oy - box_y;
    if (bx >= 0 && bx < box_w && by >= 0 && by < box_h) {
        int ix, iy, ax[4], ay[4];
        float fx, fy;
        cv_coord(bx, scale_x, ix, fx);
        cv_coord(by, scale_y, iy, fy);
        cv_cubic_coeffs(fx, ax);
        cv_cubic_coeffs(fy, ay);
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = min(max(iy - 1 + j, 0), H - 1);
            int row[3] = {0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xx = min(max(ix - 1 + i, 0), W - 1);
                const unsigned char* p = img + ((size_t)yy * W + xx) * 3;
                row[0] += ax[i] * (int)p[0]; row[1] += ax[i] * (int)p[1]; row[2] += ax[i] * (int)p[2];
            }
            acc[0] += ay[j] * row[0]; acc[1] += ay[j] * row[1]; acc[2] += ay[j] * row[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = min(max((acc[c] + (1 << 21)) >> 22, 0), 255);   // FixedPtCast<int,uchar,22>
    }
    const size_t plane = (size_t)OH * OW;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * plane + px] = (float)rgb[c] / 255.f;                // utils.py:71
}

// reference utils.py:68-71, mode='resize': cv2.resize(img, dim) (INTER_LINEAR), /255, HWC -> CHW
__global__ __launch_bounds__(256) void resize_linear_kernel(const unsigned char* __restrict__ img, int H, int W,
                                                           float* __restrict__ out, int OH, int OW,
                                                           double scale_x, double scale_y, int area2) {
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= OH * OW) return;
    const int oy = px / OW, ox = px - oy * OW;
    int rgb[3];
    if (area2) {
        const unsigned char* p = img + ((size_t)(2 * oy) * W + 2 * ox) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = ((int)p[c] + (int)p[3 + c] + (int)p[(size_t)W * 3 + c] + (int)p[(size_t)W * 3 + 3 + c] + 2) >> 2;
    } else {
        int ix, iy;
        float fx, fy;
        cv_coord(ox, scale_x, ix, fx);
        cv_coord(oy, scale_y, iy, fy);
        if (ix < 0) { fx = 0.f; ix = 0; }
        if (ix >= W - 1) { fx = 0.f; ix = W - 1; }
        const int a0 = cv_short(1.f - fx), a1 = cv_short(fx);
        const int b0 = cv_short(1.f - fy), b1 = cv_short(fy);
        const int x0 = ix, x1 = min(ix + 1, W - 1);
        const int y0 = min(max(iy, 0), H - 1), y1 = min(max(iy + 1, 0), H - 1);
        const unsigned char* r0 = img + (size_t)y0 * W * 3;
        const unsigned char* r1 = img + (size_t)y1 * W * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int s0 = (int)r0[x0 * 3 + c] * a0 + (int)r0[x1 * 3 + c] * a1;
            const int s1 = (int)r1[x0 * 3 + c] * a0 + (int)r1[x1 * 3 + c] * a1;
            rgb[c] = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
        }
    }
    const size_t plane = (size_t)OH * OW;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * plane + px] = (float)(rgb[c] & 255) / 255.f;
}

// boxes [B][cap][ld] (x1,y1,x2,y2 first), counts [B] (or NULL: every row), org_wh [B][2] = original (w,h)
__global__ void correct_boxes_kernel(const float* __restrict__ boxes, int cap, int ld, const int* __restrict__ counts,
                                     const int* __restrict__ org_wh, int img_w, int img_h, int is_letterbox, int out_xyxy,
                                     float* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = counts ? min(counts[b], cap) : cap;
    if (i >= n) return;
    const float* p = boxes + ((size_t)b * cap + i) * ld;
    float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    const int org_w = org_wh[2 * b], org_h = org_wh[2 * b + 1];
    if (((x1 + y1) + (x2 + y2)) != 0.f) {                                   // mask = labels.sum(-1) != 0
        if (is_letterbox) {                                                 // boundingbox.py:95-116
            const double rw = (double)img_w / org_w, rh = (double)img_h / org_h;
            const double ratio = rw < rh ? rw : rh;
            const int resize_w = (int)(org_w * ratio), resize_h = (int)(org_h * ratio);
            const float x_pad = (float)((img_w - resize_w) / 2), y_pad = (float)((img_h - resize_h) / 2);
            const float r = (float)ratio;
            x1 = fminf(fmaxf((x1 - x_pad) / r, 0.f), (float)org_w);
            x2 = fminf(fmaxf((x2 - x_pad) / r, 0.f), (float)org_w);
            y1 = fminf(fmaxf((y1 - y_pad) / r, 0.f), (float)org_h);
            y2 = fminf(fmaxf((y2 - y_pad) / r, 0.f), (float)org_h);
        } else {                                                            // boundingbox.py:119-137
            const float rx = (float)((double)img_w / org_w), ry = (float)((double)img_h / org_h);
            x1 = fminf(fmaxf(x1 / rx, 0.f), (float)org_w);
            x2 = fminf(fmaxf(x2 / rx, 0.f), (float)org_w);
            y1 = fminf(fmaxf(y1 / ry, 0.f), (float)org_h);
            y2 = fminf(fmaxf(y2 / ry, 0.f), (float)org_h);
        }
    }
    float* o = out + ((size_t)b * cap + i) * 4;
    o[0] = x1; o[1] = y1;
    if (out_xyxy) { o[2] = x2; o[3] = y2; }
    else { o[2] = x2 - x1; o[3] = y2 - y1; }                                // x1y1x2y2 -> xywh (boundingbox.py:10-15)
}

}  // namespace

extern "C" int yv3_letterbox(const unsigned char* img_hwc, int H, int W, float* out_chw, int out_h, int out_w, void* stream) {
    if (!img_hwc || !out_chw || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return YV3_EINVAL;
    // reference utils.py:34-42 (letterbox_transforms): python float ratio, int() truncation, // 2 offsets
    const double rw = (double)out_w / W, rh = (double)out_h / H;
    const double ratio = rw < rh ? rw : rh;
    const int box_w = (int)(W * ratio), box_h = (int)(H * ratio);
    if (box_w <= 0 || box_h <= 0) return YV3_ESHAPE;
    const int box_x = out_w / 2 - box_w / 2, box_y = out_h / 2 - box_h / 2;
    const double scale_x = 1.0 / ((double)box_w / W), scale_y = 1.0 / ((double)box_h / H);
    hipLaunchKernelGGL(letterbox_kernel, dim3(yv3_ceil_div((long long)out_h * out_w, 256)), dim3(256), 0, (hipStream_t)stream,
                       img_hwc, H, W, out_chw, out_h, out_w, box_w, box_h, box_x, box_y, scale_x, scale_y);
    YV3_CHECK_LAUNCH();
    return 0;
}

// The same kernel with the box geometry given by the caller: the evaluation pipeline's letterbox places the resized image at
// ((out_w - box_w) // 2, (out_h - box_h) // 2) (reference transforms.py:196-205, IaaLetterbox._compute_height_width_pad), which differs
// from utils.letterbox_transforms' out//2 - box//2 by one pixel when out and box have different parities; and `iaa.Scale(dim)`
// (evaluate.py:213) is the box == canvas case (a plain bicubic resize).
extern "C" int yv3_letterbox_ex(const unsigned char* img_hwc, int H, int W, float* out_chw, int out_h, int out_w,
                                int box_w, int box_h, int box_x, int box_y, void* stream) {
    if (!img_hwc || !out_chw || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return YV3_EINVAL;
    if (box_w <= 0 || box_h <= 0 || box_x < 0 || box_y < 0 || box_x + box_w > out_w || box_y + box_h > out_h) return YV3_ESHAPE;
    const double scale_x = 1.0 / ((double)box_w / W), scale_y = 1.0 / ((double)box_h / H);
    hipLaunchKernelGGL(letterbox_kernel, dim3(yv3_ceil_div((long long)out_h * out_w, 256)), dim3(256), 0, (hipStream_t)stream,
                       img_hwc, H, W, out_chw, out_h, out_w, box_w, box_h, box_x, box_y, scale_x, scale_y);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_resize_linear(const unsigned char* img_hwc, int H, int W, float* out_chw, int out_h, int out_w, void* stream) {
    if (!img_hwc || !out_chw || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return YV3_EINVAL;
    const double scale_x = 1.0 / ((double)out_w / W), scale_y = 1.0 / ((double)out_h / H);
    const int area2 = (W == 2 * out_w && H == 2 * out_h) ? 1 : 0;          // cv::resize reroutes an exact 2x2 shrink to INTER_AREA
    hipLaunchKernelGGL(resize_linear_kernel, dim3(yv3_ceil_div((long long)out_h * out_w, 256)), dim3(256), 0, (hipStream_t)stream,
                       img_hwc, H, W, out_chw, out_h, out_w, scale_x, scale_y, area2);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_correct_boxes(const float* boxes, int B, int cap, int ld, const int* counts, const int* org_wh,
                                 int img_w, int img_h, int is_letterbox, int out_xyxy, float* out_xywh, void* stream) {
    if (!boxes || !org_wh || !out_xywh || B <= 0 || cap < 0 || ld < 4 || img_w <= 0 || img_h <= 0) return YV3_EINVAL;
    if (cap == 0) return 0;
    hipLaunchKernelGGL(correct_boxes_kernel, dim3(yv3_ceil_div(cap, 128), B), dim3(128), 0, (hipStream_t)stream,
                       boxes, cap, ld, counts, org_wh, img_w, img_h, is_letterbox, out_xyxy, out_xywh);
    YV3_CHECK_LAUNCH();
    return 0;
}

// ---- stand-alone UpsampleGroup tail: nearest x2 of `up` + channel concat with `tail` (reference darknet.py:159-162), NCHW fp32.
// One thread per output element pair of a row (x even): the two outputs of an upsampled pair share one source element.
namespace {
__global__ void upsample2x_concat_kernel(const float* __restrict__ up, const float* __restrict__ tail, float* __restrict__ out,
                                         int c_up, int c_tail, int h, int w, long long pairs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // over [B][c_up + c_tail][2h][w] (pairs of columns)
    if (i >= pairs) return;
    const int W2 = 2 * w, H2 = 2 * h, C = c_up + c_tail;
    const int xp = (int)(i % w);
    long long r = i / w;
    const int y = (int)(r % H2); r /= H2;
    const int c = (int)(r % C);
    const long long b = r / C;
    float2 v;
    if (c < c_up) {
        const float s = up[((b * c_up + c) * h + (y >> 1)) * (long long)w + xp];
        v = make_float2(s, s);
    } else {
        v = *reinterpret_cast<const float2*>(tail + ((b * c_tail + (c - c_up)) * H2 + y) * (long long)W2 + 2 * xp);
    }
    *reinterpret_cast<float2*>(out + ((b * C + c) * H2 + y) * (long long)W2 + 2 * xp) = v;
}
}  // namespace

extern "C" int yv3_upsample2x_concat(const float* up, const float* tail, float* out, int B, int c_up, int c_tail, int h, int w, void* stream) {
    if (!up || (!tail && c_tail > 0) || !out || B <= 0 || c_up <= 0 || c_tail < 0 || h <= 0 || w <= 0) return YV3_EINVAL;
    const long long pairs = (long long)B * (c_up + c_tail) * (2 * h) * w;
    if (pairs > 0x7fffffffLL * 256) return YV3_ESHAPE;
    hipLaunchKernelGGL(upsample2x_concat_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, up, tail, out, c_up, c_tail, h, w, pairs);
    YV3_CHECK_LAUNCH();
    return 0;
}
