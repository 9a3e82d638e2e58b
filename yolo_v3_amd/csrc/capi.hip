// C-ABI glue: version / error strings and the dtype dispatch of the convolution entry points.
#include "yv3_common.h"

int yv3_conv2d_f32(const yv3_conv_desc* d, hipStream_t s);
int yv3_conv2d_planes(const yv3_conv_desc* d, int np, hipStream_t s);
int yv3_conv2d_f32_form(const yv3_conv_desc* d);
int yv3_conv2d_f32_launches(const yv3_conv_desc* d);      // (direct form only)
int yv3_conv2d_planes_form(const yv3_conv_desc* d, int np);

extern "C" int yv3_version(void) { return YV3_VERSION; }

extern "C" const char* yv3_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case YV3_EINVAL: return "invalid argument (null pointer or non-positive size)";
        case YV3_ESHAPE: return "shape not supported by this kernel family";
        case YV3_EWORKSPACE: return "workspace too small";
        case YV3_EDTYPE: return "unknown dtype";
        case YV3_ERCCL: return "RCCL: librccl.so not found or ncclAllGather failed";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown yv3 error";
    }
}

static int check_desc(const yv3_conv_desc* d, bool need_output = true) {
    if (!d || !d->x || !d->w || !d->beta || (need_output && !d->y && !d->dec_out)) return YV3_EINVAL;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cin <= 0 || d->cout <= 0) return YV3_EINVAL;
    if (d->k != 1 && d->k != 3) return YV3_ESHAPE;
    if (d->stride != 1 && d->stride != 2) return YV3_ESHAPE;
    if (d->k == 1 && d->stride != 1) return YV3_ESHAPE;
    if (d->cin % 32 != 0 || d->cout_pad % 32 != 0 || d->cout_pad < d->cout) return YV3_ESHAPE;
    if (d->cin_up) {
        if (d->k != 1 || !d->x2 || d->cin_up % 32 != 0 || d->cin_up >= d->cin) return YV3_ESHAPE;
        if ((d->H & 1) || (d->W & 1)) return YV3_ESHAPE;
    }
    return 0;
}

extern "C" int yv3_conv2d(const yv3_conv_desc* d, void* stream) {
    const int rc = check_desc(d);
    if (rc) return rc;
    if (d->dtype == YV3_F32) {
        if (d->out_dtype != YV3_F32) return YV3_EDTYPE;
        if (d->dec_out || !d->y) return YV3_EDTYPE;            // the fused decode lives in the plane kernels' epilogue
        return yv3_conv2d_f32(d, (hipStream_t)stream);
    }
    if (d->dtype == YV3_F32_BF16X3 || d->dtype == YV3_BF16 || d->dtype == YV3_F32_F16X2) {
        if (d->out_dtype != YV3_F32 && d->out_dtype != d->dtype) return YV3_EDTYPE;
        if (d->cin % 32) return YV3_ESHAPE;
        return yv3_conv2d_planes(d, d->dtype == YV3_BF16 ? 1 : d->dtype == YV3_F32_F16X2 ? 2 : 3, (hipStream_t)stream);
    }
    return YV3_EDTYPE;
}

extern "C" int yv3_conv2d_form(const yv3_conv_desc* d) {
    const int rc = check_desc(d, false);            // (a fused-decode head may not have its output bound yet)
    if (rc) return rc;
    if (d->dtype == YV3_F32) return yv3_conv2d_f32_form(d);
    if (d->dtype == YV3_F32_F16X2) return yv3_conv2d_planes_form(d, 2);
    if (d->dtype == YV3_F32_BF16X3 || d->dtype == YV3_BF16) return YV3_FORM_DIRECT;
    return YV3_EDTYPE;
}

extern "C" int yv3_conv2d_launches(const yv3_conv_desc* d) {
    const int form = yv3_conv2d_form(d);
    if (form < 0) return form;
    if (form != YV3_FORM_DIRECT) return 2;
    return d->dtype == YV3_F32 ? yv3_conv2d_f32_launches(d) : 1;
}

extern "C" int yv3_conv2d_sequence(const yv3_conv_desc* descs, int n, void* stream) {
    if (!descs || n < 0) return YV3_EINVAL;
    for (int i = 0; i < n; ++i) {
        const int rc = yv3_conv2d(&descs[i], stream);
        if (rc) return rc;
    }
    return 0;
}
