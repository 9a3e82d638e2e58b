// The steps immediately before and after the hot path ("next" rows 1-2):
//   letterbox_kernel      reference utils.py:34-72 (letterbox_transforms / letterbox_image / load_image):
//                         uint8 HWC RGB -> bicubic resize keeping aspect -> centred on a 128-grey canvas ->
//                         /255 -> fp32 CHW, written straight into the network's NCHW input batch
//   correct_boxes_kernel  reference boundingbox.py:95-149 (letterbox_reverse / rescale_bbox /
//                         correct_yolo_boxes): x1y1x2y2 in network pixels -> clipped xywh in the ORIGINAL image
#include "yv3_common.h"

namespace {

// cv2.INTER_CUBIC convention (reference utils.py:50): Keys kernel with A = -0.75, sample position
// fx = (dx + 0.5) * scale - 0.5, replicate border, NO antialiasing when shrinking.  cv2 itself is absent
// from this environment, so parity with its fixed-point uint8 path is unpinned (results can differ by 1
// LSB); the CPU oracle restates exactly this float formulation.
__device__ inline void cubic_w(float t, float w[4]) {
    const float A = -0.75f;
    w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

__global__ __launch_bounds__(256) void letterbox_kernel(const unsigned char* __restrict__ img, int H, int W,
                                                       float* __restrict__ out, int OH, int OW,
                                                       int box_w, int box_h, int box_x, int box_y) {
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= OH * OW) return;
    const int oy = px / OW, ox = px - oy * OW;
    float rgb[3] = {128.f, 128.f, 128.f};
    const int bx = ox - box_x, by = oy - box_y;
    if (bx >= 0 && bx < box_w && by >= 0 && by < box_h) {
        const float sx = (float)W / (float)box_w, sy = (float)H / (float)box_h;
        const float fx = ((float)bx + 0.5f) * sx - 0.5f, fy = ((float)by + 0.5f) * sy - 0.5f;
        const int ix = (int)floorf(fx), iy = (int)floorf(fy);
        float wx[4], wy[4];
        cubic_w(fx - (float)ix, wx);
        cubic_w(fy - (float)iy, wy);
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = min(max(iy - 1 + j, 0), H - 1);
            float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xx = min(max(ix - 1 + i, 0), W - 1);
                const unsigned char* p = img + ((size_t)yy * W + xx) * 3;
                row[0] += wx[i] * (float)p[0]; row[1] += wx[i] * (float)p[1]; row[2] += wx[i] * (float)p[2];
            }
            acc[0] += wy[j] * row[0]; acc[1] += wy[j] * row[1]; acc[2] += wy[j] * row[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = fminf(fmaxf(rintf(acc[c]), 0.f), 255.f);    // saturate_cast<uchar>
    }
    const size_t plane = (size_t)OH * OW;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * plane + px] = rgb[c] / 255.f;                      // utils.py:71
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
    hipLaunchKernelGGL(letterbox_kernel, dim3(yv3_ceil_div((long long)out_h * out_w, 256)), dim3(256), 0, (hipStream_t)stream,
                       img_hwc, H, W, out_chw, out_h, out_w, box_w, box_h, box_x, box_y);
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
