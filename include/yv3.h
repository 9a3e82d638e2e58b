/*
 * yv3.h -- C-ABI of libyv3.so: the MI355X (gfx950) YOLOv3 inference hot path.
 *
 * The reference (ydixon/yolo_v3) has no FFI: its boundary is the Python module surface
 * (darknet.py / yololayer.py / utils.py).  The Python package yolo_v3_amd mirrors that
 * surface and calls ONLY the functions below (via ctypes, see yolo_v3_amd/_ffi.py).  Each
 * entry point names the reference call site it replaces (file:line in /root/reference).
 *
 * Conventions
 *   - every function returns 0 on success, a negative YV3_E* code on a bad argument, or a
 *     positive hipError_t when a launch fails;
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch caching allocator) unless
 *     the name says host; nothing here allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and nothing else;
 *   - activations are NHWC ("channels last"), fp32 (dtype 0) or bf16 (dtype 1); accumulation,
 *     BN scale/shift, decode and post-processing are always fp32;
 *   - no global mutable state: concurrent calls on distinct streams/workspaces are safe.
 */
#ifndef YV3_H
#define YV3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YV3_VERSION 100          /* 0.1.0 */

#define YV3_EINVAL   (-1)        /* null pointer / non-positive dimension                  */
#define YV3_ESHAPE   (-2)        /* shape not supported by the kernel family (see function) */
#define YV3_EWORKSPACE (-3)      /* workspace too small                                     */
#define YV3_EDTYPE   (-4)        /* unknown dtype code                                      */
#define YV3_ERCCL    (-5)        /* yv3_gather_boxes: librccl not found, or ncclAllGather failed */

/* Tensor / math modes.  "Plane" tensors are NP bf16 planes [NP][B,H,W,C] (plane stride B*H*W*C).   */
#define YV3_F32  0               /* fp32 NHWC tensors, exact fp32 MFMA (v_mfma_f32_32x32x2_f32)          */
#define YV3_BF16 1               /* 1 bf16 plane, bf16 MFMA, fp32 accumulate                             */
#define YV3_F32_BF16X3 2         /* 3 bf16 planes = exact split v = p0+p1+p2 of an fp32 value; every fp32
                                    product is evaluated as its 6 leading bf16 partial products with fp32
                                    accumulation: fp32-class error at ~2.7x the fp32-MFMA rate            */
#define YV3_F32_F16X2 3          /* 2 fp16 planes hi+lo (11+11(+1 sign-carried) mantissa bits; values are
                                    saturated to +-65504, and magnitudes below 2^-14 keep an ABSOLUTE error
                                    floor of 2^-25 instead of a relative one); every fp32 product = 3 fp16
                                    MFMAs (hi*hi, hi*lo, lo*hi; the dropped lo*lo is <= 2^-22 |a*b|), fp32
                                    accumulation: ~4x fp32 epsilon per product at half the MFMA work of
                                    YV3_F32_BF16X3.  Callers should scale weights by a power of two so that
                                    max|w| is O(1) and fold the inverse into alpha (yolo_v3_amd.engine does) */

#define YV3_ACT_LINEAR 0
#define YV3_ACT_LEAKY  1         /* LeakyReLU(0.1), reference darknet.py:41 */

int yv3_version(void);
const char* yv3_error_string(int code);

/* ------------------------------------------------------------------------------------------
 * Weight preparation (one-off, at load time).  Replaces nothing on the reference's forward
 * path; it turns the parameters WeightManager loads (darknet.py:279-290) into kernel layout.
 * ------------------------------------------------------------------------------------------ */

/* OIHW fp32 [cout][cin][k][k]  ->  kernel layout for `dtype`; rows >= cout are 0.
 * YV3_F32: K-major [cout_pad][k][k][cin] fp32.  YV3_BF16 / YV3_F32_BF16X3: NP = 1 / 3 bf16 planes
 * pre-arranged tile by tile in the kernel's LDS image order; needs NP * cout_pad*k*k*cin * 2 bytes.
 * k = 1 or 3; k = 4 (plane dtypes only): the 4x4 Winograd-domain filters U = G g G^T of a 3x3 layer (yv3_conv_desc.w_wino). */
int yv3_pack_conv_weight(const float* w_oihw, void* w_packed, int cout, int cin, int k,
                         int cout_pad, int dtype, void* stream);

/* Eval-mode BatchNorm2d (darknet.py:39, eps=1e-5) as per-channel scale/shift:
 * alpha = gamma / sqrt(var + eps), beta = bias - mean * alpha. */
int yv3_fold_bn(const float* gamma, const float* bias, const float* mean, const float* var,
                float eps, float* alpha, float* beta, int channels, void* stream);

/* fp32 [n] <-> NP planes [NP][n] (np = 1 or 3: bf16, 3 is an exact, loss-free split; np = 2: fp16 hi+lo). Layout
 * conversion helpers for callers that hold fp32 tensors; not used inside the fused network plan. */
int yv3_split_planes(const float* in, void* out, long long n, int np, void* stream);
int yv3_merge_planes(const void* in, float* out, long long n, int np, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolutions.  Replace conv_bn_relu.forward (darknet.py:43-44), res_layer.forward
 * (darknet.py:52-53), the plain head Conv2d (darknet.py:118) and UpsampleGroup's
 * interpolate+cat (darknet.py:161-162).
 * ------------------------------------------------------------------------------------------ */

/* First layer, feature.mlist.0: 3 -> 32 channels, 3x3, stride 1, pad 1, + BN + leaky.
 * x is the caller's NCHW fp32 image batch [B,3,H,W] (values in [0,1]); y is NHWC [B,H,W,32] in
 * out_dtype (fp32, 1 / 3 bf16 planes, or 2 fp16 planes).
 * w_tap_major is the OIHW weight permuted to [cin][kh][kw][cout] = [27][32] fp32;
 * alpha/beta from yv3_fold_bn.
 * flags (device pointer, may be NULL): YV3_F32_F16X2 runs this layer on the fp16 matrix cores with
 * inputs scaled by 2^4 and weights by 2^8; |x| > 4094 or |w| > 255 cannot be represented and OR bit 0
 * into *flags (the same sticky status word as yv3_conv_desc.flags). */
int yv3_conv0(const float* x_nchw, const float* w_tap_major, const float* alpha, const float* beta,
              void* y_nhwc, int B, int H, int W, int out_dtype, int* flags, void* stream);

/* The first TWO layers in one launch (YV3_F32_F16X2 only): feature.mlist.0 = conv_bn_relu(3,32,3) (darknet.py:76) followed
 * by feature.mlist.1 = conv_bn_relu(32,64,3,s=2) (darknet.py:68-70), the first layer's [B,H,W,32] activation kept on chip.
 * Bit-identical to yv3_conv0(..., YV3_F32_F16X2) followed by yv3_conv2d on its output.  x_nchw as for yv3_conv0; w0 /
 * alpha0 / beta0 = the first layer's parameters as for yv3_conv0; w1_packed = the second layer's weights from
 * yv3_pack_conv_weight(cout 64, cin 32, k 3, cout_pad 64, YV3_F32_F16X2); y = 2 fp16 planes [2][B,H/2,W/2,64].
 * H and W must be multiples of 32.  flags: as yv3_conv_desc.flags. */
int yv3_conv_front(const float* x_nchw, const float* w0_tap_major, const float* alpha0, const float* beta0,
                   const void* w1_packed, const float* alpha1, const float* beta1, void* y,
                   int B, int H, int W, int* flags, void* stream);
/* The same for YV3_BF16: the first layer exactly as yv3_conv0(..., YV3_BF16) computes it (fp16 hi+lo matrix-core arithmetic, rounded
 * to bf16), w1_packed from yv3_pack_conv_weight(..., YV3_BF16), y = one bf16 plane [B,H/2,W/2,64].  Bit-identical to yv3_conv0
 * followed by yv3_conv2d in YV3_BF16. */
int yv3_conv_front_bf16(const float* x_nchw, const float* w0_tap_major, const float* alpha0, const float* beta0,
                        const void* w1_packed, const float* alpha1, const float* beta1, void* y,
                        int B, int H, int W, int* flags, void* stream);
/* The same for YV3_F32 (exact fp32; csrc/conv_front_f32.hip): the first layer as yv3_conv0(..., YV3_F32) computes it (one fma chain
 * per output over (c, kh, kw) on the vector ALUs), w1_packed [64][3][3][32] fp32 from yv3_pack_conv_weight(..., YV3_F32),
 * y fp32 NHWC [B,H/2,W/2,64].  Bit-identical to yv3_conv0 followed by yv3_conv2d in YV3_F32.  H, W multiples of 16 / 32. */
int yv3_conv_front_f32(const float* x_nchw, const float* w0_tap_major, const float* alpha0, const float* beta0,
                       const float* w1_packed, const float* alpha1, const float* beta1, float* y,
                       int B, int H, int W, void* stream);

/* The first residual block in one launch (YV3_F32_F16X2 only): feature.mlist.2 = res_layer(64) (darknet.py:46-53),
 * y = x + conv_bn_relu(32,64,3)(conv_bn_relu(64,32,1)(x)), the 32-channel intermediate kept on chip.  Bit-identical to the two
 * yv3_conv2d launches.  x, y: 2 fp16 planes [2][B,H,W,64] (H, W = the block's resolution, multiples of 16);
 * w1_packed / w2_packed from yv3_pack_conv_weight (cout_pad 32 / 64, YV3_F32_F16X2); alpha / beta from yv3_fold_bn (with the
 * weight scaling folded in, as for yv3_conv2d).  flags: as yv3_conv_desc.flags. */
int yv3_res_block64(const void* x, const void* w1_packed, const float* alpha1, const float* beta1,
                    const void* w2_packed, const float* alpha2, const float* beta2, void* y,
                    int B, int H, int W, int* flags, void* stream);
/* The same for YV3_BF16 (x, y: one bf16 plane [B,H,W,64]; weights from yv3_pack_conv_weight(..., YV3_BF16)); bit-identical to
 * the two yv3_conv2d launches in YV3_BF16. */
int yv3_res_block64_bf16(const void* x, const void* w1_packed, const float* alpha1, const float* beta1,
                         const void* w2_packed, const float* alpha2, const float* beta2, void* y,
                         int B, int H, int W, int* flags, void* stream);
/* The same for YV3_F32 (exact fp32; csrc/conv_res64_f32.hip): x, y fp32 NHWC [B,H,W,64], H a multiple of 8, W of 16 (else
 * YV3_ESHAPE: run the two launches); w1_packed [32][64], w2_packed [64][3][3][32] from yv3_pack_conv_weight(..., YV3_F32).  Same
 * products in the same order as the two yv3_conv2d launches in YV3_F32: bit-identical to them. */
int yv3_res_block64_f32(const float* x, const float* w1_packed, const float* alpha1, const float* beta1,
                        const float* w2_packed, const float* alpha2, const float* beta2, float* y,
                        int B, int H, int W, void* stream);

typedef struct yv3_conv_desc {
    const void*  x;         /* NHWC [B,H,W,cin] -- or, when cin_up > 0, the LOW-resolution map
                               [B,H/2,W/2,cin_up] that is nearest-x2 upsampled on the fly      */
    const void*  x2;        /* cin_up > 0 only: route tail NHWC [B,H,W,cin-cin_up]; else NULL  */
    const void*  w;         /* packed weights [cout_pad][k*k*cin] (yv3_pack_conv_weight)      */
    const float* alpha;     /* [cout] scale; NULL means 1.0 (plain conv)                       */
    const float* beta;      /* [cout] shift (BN) or bias (plain conv)                          */
    const void*  residual;  /* NHWC like y, added AFTER the activation (darknet.py:53); or NULL */
    void*        y;         /* NHWC [B,Ho,Wo,cout], Ho = (H + 2*pad - k)/stride + 1             */
    int B, H, W;            /* input spatial size (of the full-resolution operand)             */
    int cin;                /* total input channels (multiple of 32)                           */
    int cin_up;             /* 0, or channels taken from the upsampled `x` (they come FIRST,
                               as torch.cat((up, route_tail), 1) in darknet.py:162); k must be 1 */
    int cout, cout_pad;     /* real and padded (multiple of 32) output channels                */
    int k, stride;          /* k in {1,3}; pad = (k-1)/2 (darknet.py:34-35); stride in {1,2}   */
    int act;                /* YV3_ACT_*                                                       */
    int dtype;              /* YV3_F32 / YV3_BF16 / YV3_F32_BF16X3: format of x, x2, residual, w */
    int out_dtype;          /* format of y: == dtype, or YV3_F32 (head convs write fp32 logits) */
    int* flags;             /* optional device int32: bit 0 is OR-ed in when a YV3_F32_F16X2 output had to be
                               saturated (|value| > 65504), i.e. the fp16 planes cannot represent this
                               layer -- rerun in YV3_F32_BF16X3 or YV3_F32.  NULL: not reported.
                               Bit 1: internal scheduling error (stream-K hand-over timed out).           */
    void*  workspace;       /* optional device scratch of yv3_conv_workspace_bytes() bytes, ZERO-FILLED once by
                               the caller and then left alone: enables the persistent "stream-K" schedule of the
                               YV3_F32_F16X2 kernels (every CU gets the same number of K chunks; a tile split
                               between two workgroups is finished by the first with the accumulators the second
                               left here), used for launches of fewer than two rounds of tiles.  One workspace
                               per stream: launches that may overlap must not share it.  Opt-in because a
                               split tile is summed head + tail: results stay within the parity tolerance but
                               are no longer bit-identical for the same image at different batch positions.
                               NULL: one tile per workgroup (bitwise batch-independent).                    */
    size_t workspace_bytes;
    /* Fused YOLO decode (plane dtypes with out_dtype == YV3_F32 and cout == 3*(5+C) only): when dec_out is not
       NULL the epilogue applies yv3_decode's map to the logits and writes the result to
       dec_out + b*dec_out_batch_stride + (y*W+x)*cout + channel  (exactly what yv3_decode(logits = y, out = dec_out)
       would write, bit for bit); `y` is then optional (NULL: the logits are not materialised). */
    float*    dec_out;
    long long dec_out_batch_stride;   /* floats */
    float     dec_stride;             /* input pixels per grid cell (yololayer.py:36)            */
    float     dec_anchors[6];         /* the 3 (w,h) anchor pairs of this scale, in input pixels */
    /* Kernel-selection overrides (tuning / A-B measurements; 0 = the library's defaults).  They live in the
       descriptor -- not in environment variables or other process-global state -- so concurrent callers cannot
       affect each other. */
    unsigned  options;                /* OR of YV3_OPT_*                                          */
    int       big_tile_min;           /* plane kernels: minimum number of 256x128 tiles for which that tile is
                                         used instead of 128x128 (0: default 128 = half a round of the chip) */
    int       tune[4];                /* kernel-SELECTION experiments (0 = off; tune[0..2]: every setting gives valid results of the same
                                         arithmetic).  tune[3] is IGNORED by the shipped libyv3.so: the IO ablations it selects
                                         (invalid results) exist only in measurement builds compiled with -DYV3_MEASURE */
    /* Plane strides in ELEMENTS for plane dtypes, 0 = the packed default B*H*W*C of that tensor.  They let one launch work
       on a batch SLICE of [NP][B_total,H,W,C] tensors (x / x2 / y + residual: base pointers offset by b0*H*W*C, B = slice
       size, strides = those of the full tensors): the engine runs a layer whose tiles fill between one and two rounds of the
       chip as "exactly one round" + "the rest" (bit-identical results: the K order does not depend on the tiling). */
    long long x_plane_stride, x2_plane_stride, y_plane_stride;
    /* Winograd F(2x2,3x3) path of the YV3_F32_F16X2 kernels (k = 3, stride 1, cin_up = 0, out_dtype == dtype; csrc/winograd.hip):
       when w_wino is not NULL the layer runs as  V = B^T d B (one streaming launch: 4x4 input tiles at stride 2, transformed in
       fp32, x 1/4, split hi/lo, written to wino_ws as [2][16][T][cin] fp16 planes, T = B*ceil(H/2)*ceil(W/2))  ->  sixteen
       T x cout x cin GEMMs on the matrix cores, folded on the fly into the four outputs of every tile (Y = A^T M A), same
       epilogue as the direct kernel: 2.25x fewer matrix instructions per output.  w_wino = the 4x4 transformed filters
       U = G g G^T packed with yv3_pack_conv_weight(k = 4); alpha_wino = alpha with U's per-row power-of-two scale and the
       x 4 of the input scaling folded in.  Results differ from the direct kernel by fp32 round-off only (per-layer error
       ~2x the direct scheme's, tools/winograd_numerics.py).  wino_ws: scratch of yv3_wino_workspace_bytes(B,H,W,cin) bytes,
       ZERO-FILLED once by the caller and then left alone (its tail holds the hand-over flags of the even schedule);
       launches that may overlap must not share it.  The library takes this path by the layer's count of 128x128 Winograd tiles
       (r = tiles / CUs: r >= 0.62 up to one round, r / ceil(r) >= 0.75 beyond -- measured crossovers; r >= 0.27 with
       YV3_OPT_TWO_LANES; YV3_OPT_WINO_ALWAYS: whenever w_wino is set) and the direct
       kernel otherwise -- the choice depends on B, so the same image may be computed by either form at different batch sizes
       (both within fp32 round-off of the exact result, not bit-identical to each other).  YV3_OPT_WINO_EVEN: stream-K schedule
       over transform positions (one persistent workgroup per CU; a split tile is summed head + tail). */
    const void*  w_wino;
    const float* alpha_wino;
    void*        wino_ws;
    size_t       wino_ws_bytes;
    /* Winograd F(4x4,3x3) path of the YV3_F32 kernels (k = 3, stride 1, cin_up = 0, cout % 64 == 0, cin == 64 or cin % 128 == 0; csrc/conv_wino4_f32.hip): when
       w_wino4 is not NULL the layer may run as  V = B^T d B over 6x6 input patches at stride 4 (points 0, 1, -1, 1/2, -2, inf; fp32;
       written to wino_ws as [36][T][cin], T = B*ceil(H/4)*ceil(W/4): yv3_wino4_workspace_bytes)  ->  thirty-six T x cout x cin GEMMs on
       v_mfma_f32_16x16x4_f32, folded on the fly into the sixteen outputs of every tile: 4x fewer matrix instructions than the direct
       form, 1.78x fewer than F(2x2,3x3).  w_wino4 = yv3_pack_wino4_weight_f32 of U = G g G^T; scale / shift = alpha / beta.  Results
       differ from the direct kernel by fp32 round-off (whole network: within 1.4x of the direct form's distance from an fp64
       evaluation on hostile data, equal on the headline data -- tools/winograd_f32_gate.py).  The library takes this form when its
       64-channel x 32-tile workgroups fill the chip (yv3_conv2d_form == YV3_FORM_WINOGRAD4), else F(2x2) by w_wino's rule, else the
       direct kernel.  Schedule: one (32 tiles x 64 channels) item per workgroup while the items fill whole rounds of the chip (two
       workgroups per CU); the items of a last, partial round are dealt out by patch row over all the slots -- an item split between
       workgroups of one XCD is summed part by part in workgroup order through the hand-over area at the END of wino_ws (the last
       yv3_wino4_workspace_bytes - V bytes: ZERO-FILLED once by the caller, then left alone; one wino_ws per stream).  Which items are
       split depends on B: the same image is then summed in a different (fixed per shape) order at different batch sizes / positions;
       YV3_OPT_WINO4_TILES keeps one item per workgroup.  A hand-over that times out sets bit 1 of *flags. */
    const void*  w_wino4;
} yv3_conv_desc;

#define YV3_OPT_NO_PINGPONG 1u    /* fp16-plane 8-wave tiles: single-phase main loop instead of the two-group ping-pong */
#define YV3_OPT_K3S1        2u    /* 3x3 stride-1 plane convs: the kw-tap-reuse kernel (conv_planes_k3s1.hip)          */
#define YV3_OPT_WINO_EVEN   4u    /* Winograd stage: even (stream-K over transform positions) schedule instead of one tile per workgroup */
#define YV3_OPT_WINO_ALWAYS 8u    /* Winograd stage whenever w_wino is set, whatever the tile count                                     */
#define YV3_OPT_TWO_LANES   16u   /* the caller runs an equal launch sequence on a second stream at the same time (two lanes of one batch):
                                     the Winograd rule's lower bound counts both lanes' tiles (0.27 instead of 0.55 rounds per launch)  */
#define YV3_OPT_WINO4_TILES 32u   /* F(4x4,3x3) stage: one item per workgroup only -- no even schedule (whose ranges wait for their partner
                                     workgroups: callers that share the GPU with other work; results then do not depend on the batch size) */
#define YV3_OPT_TILE_SHIFT  8     /* bits 8..15: force a tile configuration of the fp16-plane kernels (0 = automatic):
                                     1 = 256x128 / 8 waves, 2 = 128x128 / 8 waves, 3 = 128x128 / 4 waves, two workgroups per CU */

/* Size of yv3_conv_desc.workspace. */
size_t yv3_conv_workspace_bytes(void);

/* Size of yv3_conv_desc.wino_ws for a B x H x W x cin input (YV3_F32_F16X2). */
size_t yv3_wino_workspace_bytes(int B, int H, int W, int cin);

/* Size of yv3_conv_desc.wino_ws that the F(4x4,3x3) form of a YV3_F32 layer needs: V + the hand-over area (never more than
 * yv3_wino_workspace_bytes for H, W >= 4).  U [cout][cin][6][6] fp32 = G g G^T (points 0, 1, -1, 1/2, -2, inf; computed by the caller, in fp64 and rounded once)
 * -> the GEMM stage's packed image of cout*cin*36 floats; cout % 64 == 0, cin == 64 or a multiple of 128. */
size_t yv3_wino4_workspace_bytes(int B, int H, int W, int cin);
int yv3_pack_wino4_weight_f32(const float* u_oc66, float* packed, int cout, int cin, void* stream);

/* y = act(conv(x) * alpha + beta) (+ residual), implicit GEMM on the MFMA units. */
int yv3_conv2d(const yv3_conv_desc* desc, void* stream);

/* Which form would yv3_conv2d run for this descriptor ON THE CURRENT DEVICE (the per-launch rule depends on the CU count)?
 * YV3_FORM_DIRECT (0): the direct implicit-GEMM kernel (36 multiplications per 2x2 outputs and channel pair for a 3x3 layer);
 * YV3_FORM_WINOGRAD (1): input transform + 16-position GEMM (16 per 2x2 outputs: 2.25x fewer matrix instructions).
 * Negative: the YV3_E* code yv3_conv2d would return for it.  Launches nothing; used by tests (assert which plan ran) and by
 * bench.py (executed vs algorithmic FLOPs). */
#define YV3_FORM_DIRECT   0
#define YV3_FORM_WINOGRAD 1
#define YV3_FORM_WINOGRAD4 2     /* YV3_F32: F(4x4,3x3), 36 multiplications per 4x4 outputs and channel pair (4x fewer than direct) */
int yv3_conv2d_form(const yv3_conv_desc* desc);

/* How many kernels does yv3_conv2d launch for this descriptor on the current device?  1; 2 for the Winograd forms (input transform + GEMM
 * stage) and for a YV3_F32 1x1 layer whose whole rounds of tiles run on the persistent GEMM (csrc/conv_gemm_f32.hip) and the rest on the
 * small tiles.  Negative: the YV3_E* code yv3_conv2d would return.  Launches nothing; profiling tools use it to map a kernel trace to layers. */
int yv3_conv2d_launches(const yv3_conv_desc* desc);

/* Run `n` convolutions back to back on `stream` (one host call for a whole network plan). */
int yv3_conv2d_sequence(const yv3_conv_desc* descs, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * YOLO head decode.  Replaces YoloLayer.forward, inference branch (yololayer.py:31-59,97-105).
 *   bx=(sigmoid(tx)+gx)*stride  by=(sigmoid(ty)+gy)*stride
 *   bw=exp(tw)*(aw/stride)*stride  bh likewise   conf=sigmoid(to)  cls_k=sigmoid(t_k)
 * out row = (y*W + x)*3 + a (yololayer.py:104); columns cx,cy,w,h,conf,cls0..
 * ------------------------------------------------------------------------------------------ */

/* logits NHWC [B,H,W,3*(5+C)] with row pitch `ld_logits` floats per pixel (>= 3*(5+C)).
 * out[b] starts at out + b*out_batch_stride (floats) and holds H*W*3 rows of (5+C) floats:
 * pass out = base + row_offset*(5+C) to write one scale into a concatenated [B,N,5+C] tensor.
 * anchors: 3 (w,h) pairs in input pixels (host pointer, copied by value). */
int yv3_decode(const float* logits, int ld_logits, const float* anchors_host, float stride,
               float* out, long long out_batch_stride, int B, int H, int W, int num_class,
               void* stream);

/* Same, reading the reference's NCHW layout [B,3*(5+C),H,W] (channel = a*(5+C)+attr,
 * yololayer.py:42) -- the drop-in YoloLayer.forward(x) entry. */
int yv3_decode_nchw(const float* logits_nchw, const float* anchors_host, float stride,
                    float* out, long long out_batch_stride, int B, int H, int W, int num_class,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Geometry helpers.  Replace boundingbox.py:25-29, utils.py:98-119, utils.py:122-146.
 * ------------------------------------------------------------------------------------------ */

/* boxes [n,4] cxcywh -> x1y1x2y2 (in place allowed: out may equal in). */
int yv3_cxcywh_to_xyxy(const float* in, float* out, long long n, void* stream);

/* out[n1,n2] = IOU(b1[i], b2[j]); boxes have `ld` floats per row (>= 4);
 * mode 0 = x1y1x2y2, 1 = cxcywh.  No +1, no epsilon: 0/0 gives NaN like the reference. */
int yv3_iou_matrix(const float* b1, int n1, int ld1, const float* b2, int n2, int ld2,
                   int mode, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Post-processing.  Replaces utils.postprocessing (utils.py:226-258) with
 * get_nms_detections (utils.py:148-202) / get_raw_detections (utils.py:204-224).
 *
 * Two calls so that the caller can size buffers from the candidate counts if it wants to:
 *   1. yv3_postproc_filter : score = cls*conf, confidence filter, candidate compaction
 *   2. yv3_postproc_nms    : order by (class asc, score desc, row asc), IOU bit masks,
 *                            greedy scan, compaction into [B,cap,7]
 * ------------------------------------------------------------------------------------------ */

/* Bytes of candidate storage for up to `max_cand` candidates per image. */
size_t yv3_postproc_cand_bytes(int B, int max_cand, int num_class);

/* Bytes of scratch yv3_postproc_nms needs for up to `max_n` candidates per image. */
size_t yv3_postproc_nms_workspace_bytes(int B, int max_n, int num_class);

#define YV3_PP_EVAL 1   /* is_eval=True: every (row,k) with cls_k*conf > thr (utils.py:238)          */
#define YV3_PP_PROB 2   /* caller guarantees 0 <= cls_k <= 1 (sigmoid outputs): rows with conf <= thr
                           cannot pass and are skipped unread.  Same result, ~5x less traffic.      */

/* dets: decoded detections [B,N,5+C] (cx,cy,w,h,conf,cls..), row pitch 5+C, NOT modified.
 * mode: OR of YV3_PP_*.  Without YV3_PP_EVAL: one candidate per row whose max_k(cls_k*conf) > thr,
 * class = first argmax (utils.py:242-246).
 * cand: opaque buffer of yv3_postproc_cand_bytes(B,max_cand,C); cand_counts [B] int32 receives
 * the number of candidates found per image (may exceed max_cand: then the excess was dropped
 * and the caller must retry with a larger buffer). */
int yv3_postproc_filter(const float* dets, int B, int N, int num_class, float conf_thr,
                        int mode, void* cand, int max_cand, int* cand_counts, void* stream);

/* max_n: upper bound on candidates per image that is actually processed (<= max_cand; images
 * with more candidates are truncated).  It sizes the workspace, so a caller that has read
 * cand_counts back can pass max(cand_counts); a sync-free caller passes max_cand.
 * out_boxes [B,cap,7] = x1,y1,x2,y2,conf,score,cls per kept box, ordered by class ascending then
 * score descending (utils.py:161-172,193-199); out_counts [B] = number kept (may exceed cap: rows
 * beyond cap are dropped).  use_nms = 0 reproduces get_raw_detections: every candidate, row order.
 * Zero-area / inverted boxes are dropped and never suppress (self-IOU not > thr, utils.py:182). */
int yv3_postproc_nms(const float* dets, int B, int N, int num_class, float nms_thr, int use_nms,
                     const void* cand, int max_cand, const int* cand_counts, int max_n,
                     float* out_boxes, int cap, int* out_counts,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Neighbours of the hot path: input preparation and box un-mapping.
 * ------------------------------------------------------------------------------------------ */

/* Replaces utils.letterbox_image + the /255, HWC->CHW of utils.load_image (utils.py:44-72):
 * img_hwc uint8 RGB [H,W,3] (device) -> out_chw fp32 [3,out_h,out_w] in [0,1]: cv2.resize(INTER_CUBIC) -- OpenCV's
 * fixed-point 8-bit path: A=-0.75 coefficients as 11-bit shorts, int32 horizontal pass, (sum + 2^21) >> 22 vertical
 * pass, no antialias -- to box = int(size*min(out_w/W,out_h/H)), centred (offset out/2 - box/2, utils.py:34-42) on
 * a 128-grey canvas.  Pass out_chw = batch + b*3*out_h*out_w.
 * PARITY WITH cv2 IS UNPINNED: cv2 is absent from the build image, so the kernel is checked bit for bit against a
 * restatement of OpenCV's scalar fixed-point path (oracle_cpu.cv_resize_cubic_u8), not against cv2 output.  SIMD builds
 * of OpenCV run the vertical cubic pass in float32 (VResizeCubicVec_32s8u) and can differ from the scalar formula by
 * 1 LSB on ~1e-4 of the pixels: expect agreement with a given cv2 build to within 1 LSB (1/255 of the input), unverified. */
int yv3_letterbox(const unsigned char* img_hwc, int H, int W, float* out_chw, int out_h, int out_w,
                  void* stream);

/* The same resampling with the box geometry GIVEN (box_w x box_h pixels of the resized image at (box_x, box_y) on the 128-grey
 * canvas): the evaluation pipeline's letterbox (transforms.py:144-209, IaaLetterbox: box at ((out_w-box_w)//2, (out_h-box_h)//2) --
 * one pixel off utils.letterbox_transforms' out//2 - box//2 when the parities differ) and its plain `iaa.Scale(dim)` (evaluate.py:213:
 * box == canvas, bicubic).  Same cv2 caveat as yv3_letterbox.  YV3_ESHAPE if the box does not fit the canvas. */
int yv3_letterbox_ex(const unsigned char* img_hwc, int H, int W, float* out_chw, int out_h, int out_w,
                     int box_w, int box_h, int box_x, int box_y, void* stream);

/* Replaces load_image(mode='resize') (utils.py:68-71): cv2.resize(img, (out_w,out_h)) with the default INTER_LINEAR
 * (OpenCV's fixed-point 8-bit path; an exact 2x2 shrink is INTER_AREA, as in cv::resize), /255, HWC -> CHW. */
int yv3_resize_linear(const unsigned char* img_hwc, int H, int W, float* out_chw, int out_h, int out_w,
                      void* stream);

/* Replaces boundingbox.correct_yolo_boxes (boundingbox.py:139-149) = letterbox_reverse (:95-116) or
 * rescale_bbox (:119-137) followed by x1y1x2y2 -> xywh.  boxes [B][cap][ld] with x1,y1,x2,y2 in the
 * first four columns (e.g. the [B,cap,7] output of yv3_postproc_nms, ld = 7); counts [B] valid rows per
 * image (NULL: all cap rows); org_wh [B][2] int32 original (width,height) per image (device).
 * out [B][cap][4] in original-image pixels, clipped to the image like the reference: x,y,w,h
 * (out_xyxy = 0, what correct_yolo_boxes returns) or the intermediate x1,y1,x2,y2 (out_xyxy = 1,
 * what letterbox_reverse / rescale_bbox return). */
int yv3_correct_boxes(const float* boxes, int B, int cap, int ld, const int* counts, const int* org_wh,
                      int img_w, int img_h, int is_letterbox, int out_xyxy, float* out, void* stream);

/* Stand-alone UpsampleGroup tail (reference darknet.py:159-162: ``F.interpolate(out, scale_factor=2, mode='nearest')`` then
 * ``torch.cat((out, route_tail), 1)``), NCHW fp32:  out[b, c, y, x] = up[b, c, y/2, x/2] for c < c_up, tail[b, c - c_up, y, x] beyond.
 * up [B, c_up, h, w], tail [B, c_tail, 2h, 2w], out [B, c_up + c_tail, 2h, 2w].  Pure data movement (bit-exact).  Inside YoloNet
 * this never runs: the consumer convolution gathers both sources itself (yv3_conv_desc.x / x2 / cin_up). */
int yv3_upsample2x_concat(const float* up, const float* tail, float* out, int B, int c_up, int c_tail, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU ("yv3_gather_boxes(comm, ...)").  The reference is single-GPU; images are independent
 * (utils.py:152), so the sharded path has exactly ONE exchange: an all-gather of every rank's [b_local, rows, 7] fp32 payload --
 * rows = cap + 1: the [b_local, cap, 7] output of yv3_postproc_nms plus one row per image carrying its int32 candidate count,
 * kept count and the status word (yv3_conv_desc.flags), bit-cast; layout = yolo_v3_amd/dist.py:pack_payload / unpack_payload --
 * into gathered [world * b_local, rows, 7], rank order.
 *   rccl_comm: an ncclComm_t owned by the caller (ncclCommInitRank ...); stream: hipStream_t the collective is enqueued on.
 * One ncclAllGather over RCCL/xGMI, nothing else; no allocation, no synchronisation.  libyv3.so does not link librccl: the symbol is
 * resolved at the first call (from the process if RCCL is already loaded, else librccl.so.1 / librccl.so), YV3_ERCCL if absent.
 * The Python product (`detect_sharded`) issues the same collective through torch.distributed (backend "nccl" = RCCL), whose
 * communicator cannot be handed out; this entry point is for hosts that own their communicator.
 * ------------------------------------------------------------------------------------------ */
int yv3_gather_boxes(const float* payload, float* gathered, int b_local, int rows, void* rccl_comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YV3_H */
