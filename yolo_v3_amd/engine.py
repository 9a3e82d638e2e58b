"""Execution plan for the YOLOv3 hot path on one MI355X.

The reference runs ``YoloNet.forward`` (darknet.py:198-231) as ~230 eager ATen calls on NCHW
tensors.  Here the whole network is a static plan:

* weights are packed once into the K-major layout of the implicit-GEMM kernels and BatchNorm
  (eval) is folded to a per-channel scale/shift applied in the conv epilogue;
* activations are NHWC buffers allocated once per (batch, H, W) -- 288 GB of HBM3E per GPU means
  no buffer juggling is needed even at batch 256;
* one ``yv3_conv0`` call (reads the caller's NCHW image batch directly) + ONE
  ``yv3_conv2d_sequence`` call for the other 74 convolutions (residual adds, the plain head
  convs and the upsample+concat are all epilogue / gather variants of the same kernel) + three
  ``yv3_decode`` calls that write straight into the concatenated ``[B, N, 5+C]`` tensor, whose
  row order equals the reference's ``torch.cat((det1, det2, det3), 1)``.
"""
import ctypes
import os

import torch

from . import _ffi, arch

from ._ffi import ConvDesc, F32, BF16, F32X3, F32H2, ACT_LEAKY, ACT_LINEAR

_TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F32X3: torch.bfloat16, F32H2: torch.float16}   # element type of activation storage
PLANES = {F32: 0, BF16: 1, F32X3: 3, F32H2: 2}   # 0: plain fp32 NHWC tensor; n > 0: n 16-bit planes [n][B,H,W,C]


def alloc_act(B, h, w, c, dtype, device):
    """Activation buffer in the layout of `dtype` (include/yv3.h): fp32 NHWC, or NP bf16 planes."""
    np_ = PLANES[dtype]
    shape = (B, h, w, c) if np_ == 0 else (np_, B, h, w, c)
    return torch.empty(shape, device=device, dtype=_TORCH_DTYPE[dtype])


def to_planes(x_nhwc_f32, dtype):
    """fp32 NHWC tensor -> the activation layout of `dtype` (exact for F32X3)."""
    np_ = PLANES[dtype]
    if np_ == 0:
        return x_nhwc_f32.float().contiguous()
    x = x_nhwc_f32.float().contiguous()
    out = torch.empty((np_,) + tuple(x.shape), device=x.device, dtype=_TORCH_DTYPE[dtype])
    _ffi.check(_ffi.lib().yv3_split_planes(x.data_ptr(), out.data_ptr(), x.numel(), np_, _ffi.stream_ptr()), "yv3_split_planes")
    return out


def from_planes(t, dtype):
    """Activation buffer of `dtype` -> fp32 NHWC tensor."""
    np_ = PLANES[dtype]
    if np_ == 0 or t.dtype == torch.float32:
        return t
    out = torch.empty(tuple(t.shape[1:]), device=t.device, dtype=torch.float32)
    _ffi.check(_ffi.lib().yv3_merge_planes(t.data_ptr(), out.data_ptr(), out.numel(), np_, _ffi.stream_ptr()), "yv3_merge_planes")
    return out


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class StreamKTimeout(_ffi.Yv3Error):
    """The kernels' status bit 1: a stream-K hand-over did not arrive (see Engine.raise_if_overflowed)."""


class RangeOverflow(_ffi.Yv3Error):
    """The kernels' status bit 0: a value left the fp16 range of the F32H2 planes (or BF16's first layer) and was saturated."""


class PackedConv:
    """Device-side parameters of one convolution in kernel layout."""
    __slots__ = ("spec", "w", "alpha", "beta", "cout_pad", "w_wino", "alpha_wino", "w_wino4")

    def __init__(self, spec, w, alpha, beta, cout_pad, w_wino=None, alpha_wino=None, w_wino4=None):
        self.spec, self.w, self.alpha, self.beta, self.cout_pad = spec, w, alpha, beta, cout_pad
        self.w_wino, self.alpha_wino = w_wino, alpha_wino      # Winograd-domain filters of an eligible 3x3 layer (F32H2, F32)
        self.w_wino4 = w_wino4                                 # F(4x4,3x3) filters (exact-fp32 mode, csrc/conv_wino4_f32.hip)


def conv_params(module):
    """(weight, bn or None, bias or None) of a conv_bn_relu container or a plain nn.Conv2d."""
    if isinstance(module, torch.nn.Conv2d):
        return module.weight, None, module.bias
    return module.conv.weight, module.bn, None


def measure_env(name, default=None):
    """Tuning / A-B overrides from the environment are honoured ONLY in measurement sessions (YV3_MEASURE=1, set by tools/gpu.sh): in
    normal use a stray YV3_* variable changes nothing -- kernel selection is a function of the descriptor, the product knobs are
    attributes of the net (net.math_mode, .stream_k, .winograd, .deterministic, .lanes, ...)."""
    if os.environ.get("YV3_MEASURE") != "1":
        return default
    return os.environ.get(name, default)


WINO_MIN_CIN = int(measure_env("YV3_WINO_MIN_CIN", "256") or 256)     # Winograd F(2x2,3x3) for 3x3 / stride-1 layers with at least this many input channels (26x26 and 13x13 at
                       # 416x416): below, the transformed input (16 B per input element through HBM) costs more than the MFMAs saved
_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


SK_AUTO_CELLS = int(measure_env("YV3_SK_AUTO_CELLS", "1536") or 1536)     # stream-K by default up to this many 32x32 cells per batch (Plan.__init__)


WINO_MIN_CIN_F32 = 64  # exact-fp32 mode: fp32 MFMA runs at the vector rate, every 3x3 layer is matrix-bound -> from the 104x104 layers down


def wino_eligible(spec, dtype):
    if not (spec.k == 3 and spec.stride == 1 and spec.bn):
        return False
    if dtype == F32H2:
        return spec.cin >= WINO_MIN_CIN
    return dtype == F32 and spec.cin >= WINO_MIN_CIN_F32 and spec.cout % 128 == 0


def pack_wino(weight_f32, alpha_bn, spec, cout_pad, dtype=F32H2):
    """U = G g G^T (fp64 -> fp32) of every 3x3 filter, packed as a 16-tap ("k = 4") filter bank.  F32H2: per-output-channel
    power-of-two scaling as for the direct weights, alpha_wino = alpha * 2^-e * 4 (the input transform carries a factor
    1/4).  F32: plain fp32 U, alpha_wino = alpha."""
    lib = _ffi.lib()
    dev = weight_f32.device
    # U[o,c,i,l] = sum_jk G[i,j] g[o,c,j,k] G[l,k] in fp64, as 24 scaled adds (elementwise torch ops: no vendor GEMM anywhere)
    g64 = weight_f32.double()
    t = torch.stack([sum(_WINO_G[i][j] * g64[:, :, j, :] for j in range(3) if _WINO_G[i][j] != 0.0) for i in range(4)], 2)      # [O,C,4,3]
    U = torch.stack([sum(_WINO_G[l][k] * t[:, :, :, k] for k in range(3) if _WINO_G[l][k] != 0.0) for l in range(4)], 3)        # [O,C,4,4]
    U = U.float().contiguous()
    if dtype == F32:
        wp = torch.empty(cout_pad * 16 * spec.cin, device=dev, dtype=torch.float32)
        _ffi.check(lib.yv3_pack_conv_weight(U.data_ptr(), wp.data_ptr(), spec.cout, spec.cin, 4, cout_pad, F32, _ffi.stream_ptr()),
                   "yv3_pack_conv_weight(k=4)")
        return wp, alpha_bn.clone()
    umax = U.abs().amax(dim=(1, 2, 3))
    e = torch.where(umax > 0, -torch.floor(torch.log2(umax.clamp(min=1e-38))), torch.zeros_like(umax)).clamp(-100.0, 100.0)
    U = (U * torch.exp2(e).view(-1, 1, 1, 1)).contiguous()
    alpha_w = (alpha_bn * torch.exp2(-e) * 4.0).contiguous()
    wp = torch.empty(2 * cout_pad * 16 * spec.cin, device=dev, dtype=torch.float16)
    _ffi.check(lib.yv3_pack_conv_weight(U.data_ptr(), wp.data_ptr(), spec.cout, spec.cin, 4, cout_pad, F32H2, _ffi.stream_ptr()),
               "yv3_pack_conv_weight(k=4)")
    return wp, alpha_w


# F(4x4,3x3) with the Toom-Cook points (0, 1, -1, 1/2, -2, inf): G [6][3] (csrc/conv_wino4_f32.hip holds B^T and A^T; the point set is the
# one whose whole-network error equals the direct form's, tools/winograd_f32_gate.py)
_WINO4_G = ((1.0, 0.0, 0.0), (1.0 / 3, 1.0 / 3, 1.0 / 3), (-1.0 / 3, 1.0 / 3, -1.0 / 3),
            (-16.0 / 15, -8.0 / 15, -4.0 / 15), (1.0 / 15, -2.0 / 15, 4.0 / 15), (0.0, 0.0, 1.0))


def wino4_filters(weight_f32):
    """U = G g G^T [O,C,6,6] of a 3x3 filter bank: fp64 elementwise torch ops (no vendor GEMM), rounded to fp32 once."""
    g64 = weight_f32.double()
    t = torch.stack([sum(_WINO4_G[i][j] * g64[:, :, j, :] for j in range(3) if _WINO4_G[i][j] != 0.0) for i in range(6)], 2)      # [O,C,6,3]
    U = torch.stack([sum(_WINO4_G[l][k] * t[:, :, :, k] for k in range(3) if _WINO4_G[l][k] != 0.0) for l in range(6)], 3)        # [O,C,6,6]
    return U.float().contiguous()


def pack_wino4(weight_f32, spec):
    """The exact-fp32 mode's F(4x4,3x3) filter image (yv3_pack_wino4_weight_f32); scale / shift stay the layer's alpha / beta."""
    U = wino4_filters(weight_f32)
    wp = torch.empty(spec.cout * spec.cin * 36, device=weight_f32.device, dtype=torch.float32)
    _ffi.check(_ffi.lib().yv3_pack_wino4_weight_f32(U.data_ptr(), wp.data_ptr(), spec.cout, spec.cin, _ffi.stream_ptr()), "yv3_pack_wino4_weight_f32")
    return wp


def pack_conv(module, spec, dtype, winograd=False, winograd4=True):
    """Pack one conv (+BN) for the HIP kernels.  Parameters must already be on the GPU."""
    lib = _ffi.lib()
    weight, bn, bias = conv_params(module)
    _ffi.require_cuda(weight, "parameter %s" % spec.name)
    dev = weight.device
    s = _ffi.stream_ptr()
    w32 = weight.detach().float().contiguous()
    if bn is not None:
        alpha = torch.empty(spec.cout, device=dev, dtype=torch.float32)
        beta = torch.empty(spec.cout, device=dev, dtype=torch.float32)
        g, b = bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous()
        m, v = bn.running_mean.detach().float().contiguous(), bn.running_var.detach().float().contiguous()
        _ffi.check(lib.yv3_fold_bn(g.data_ptr(), b.data_ptr(), m.data_ptr(), v.data_ptr(), float(bn.eps),
                                   alpha.data_ptr(), beta.data_ptr(), spec.cout, s), "yv3_fold_bn")
    else:
        alpha = None
        beta = bias.detach().float().contiguous().clone()
    w_orig, alpha_bn = w32, alpha
    if dtype == F32H2 and spec.cin != 3:
        # fp16 planes keep a relative precision of 2^-22 only while the `lo` part is a normal fp16 number, i.e. for
        # |w| >= 2^-3 after scaling: bring EVERY OUTPUT CHANNEL's weights to max|w_row| in [1,2) with its own exact
        # power-of-two scale and fold the inverse into that channel's epilogue scale (also exact) -- rows whose
        # weights are all tiny relative to the layer maximum then keep their relative precision
        wmax = w32.abs().amax(dim=(1, 2, 3))
        e = torch.where(wmax > 0, -torch.floor(torch.log2(wmax.clamp(min=1e-38))), torch.zeros_like(wmax))
        e = e.clamp(-100.0, 100.0)
        w32 = w32 * torch.exp2(e).view(-1, 1, 1, 1)
        alpha = (alpha if alpha is not None else torch.ones(spec.cout, device=dev, dtype=torch.float32)) * torch.exp2(-e)
    if spec.cin == 3:
        # first layer: direct-conv kernel wants [cin][kh][kw][cout] fp32
        return PackedConv(spec, w32.permute(1, 2, 3, 0).contiguous(), alpha, beta, spec.cout)
    cout_pad = (spec.cout + 31) // 32 * 32
    if dtype != F32 and cout_pad > 128:
        cout_pad = (cout_pad + 127) // 128 * 128     # plane kernels tile the channels by 128 (any class count, e.g. 3*(5+40) = 135 -> 256)
    nw = cout_pad * spec.k * spec.k * spec.cin
    wp = torch.empty(max(1, PLANES[dtype]) * nw, device=dev, dtype=_TORCH_DTYPE[dtype])
    _ffi.check(lib.yv3_pack_conv_weight(w32.data_ptr(), wp.data_ptr(), spec.cout, spec.cin, spec.k,
                                        cout_pad, dtype, s), "yv3_pack_conv_weight")
    if winograd and wino_eligible(spec, dtype):
        ww, aw = pack_wino(w_orig, alpha_bn, spec, cout_pad, dtype)
        w4 = pack_wino4(w_orig, spec) if (dtype == F32 and winograd4 and spec.cout % 64 == 0 and (spec.cin == 64 or spec.cin % 128 == 0) and cout_pad == spec.cout) else None
        return PackedConv(spec, wp, alpha, beta, cout_pad, ww, aw, w4)
    return PackedConv(spec, wp, alpha, beta, cout_pad)


def tuning_options():
    """Kernel-selection overrides for A/B measurements (tools/): read from the environment HERE, on the host side of
    the C-ABI, and passed in yv3_conv_desc.options / .big_tile_min -- the library itself reads no environment.  Only with
    YV3_MEASURE=1 (measure_env)."""
    opts = 0
    if measure_env("YV3_NO_PP"):
        opts |= _ffi.OPT_NO_PINGPONG
    if measure_env("YV3_K3S1"):
        opts |= _ffi.OPT_K3S1
    if measure_env("YV3_WINO_EVEN"):
        opts |= _ffi.OPT_WINO_EVEN
    if measure_env("YV3_WINO_ALWAYS"):
        opts |= _ffi.OPT_WINO_ALWAYS
    opts |= (int(measure_env("YV3_TILE", "0") or 0) & 0xff) << 8
    return opts, int(measure_env("YV3_BIG_MIN", "0") or 0)


def batch_split(B, ho, wo, cout_pad, ncu):
    """(B0, B1) when a 3x3 plane-kernel launch should run as two: its 256x128 tiles fill between one and two rounds of the
    chip's `ncu` CUs with a partial second round of 15-60 % -- then the first B0 images fill (at most) exactly one round and the
    remaining B1 run on their own.  Opt-in (`net.batch_split = True`): in isolation the 13x13 layers at bs=64 (344 tiles =
    1.34 rounds) take 0.349 ms as one launch and 0.219 + 0.106 ms as two (profiles/r02_batch_split_probe.log), but inside the
    network the step time is unchanged (13.21 vs 13.21 ms, same box, alternating runs) -- the chip is power-limited, a
    partially filled round simply clocks higher.  Results are bit-identical (the K order does not depend on the tiling).
    None: one launch."""
    if cout_pad % 128:
        return None
    ntn = cout_pad // 128

    def tiles(b):
        return -(-(b * ho * wo) // 256) * ntn
    t = tiles(B)
    if not (ncu < t < 2 * ncu) or not (0.15 <= t / ncu - 1.0 <= 0.6):
        return None
    fit = [b for b in range(1, B) if tiles(b) <= ncu]
    if not fit:
        return None
    return fit[-1], B - fit[-1]


def make_desc(pc, x, y, B, H, W, residual=None, x2=None, cin_up=0, dtype=F32, out_dtype=None, flags=None, workspace=None,
              batch=None, wino_ws=None, two_lanes=False, wino_always=False, wino4_tiles=False):
    """`batch` = (b0, nb): the descriptor covers images b0 .. b0+nb-1 of the [NP][B,...] plane tensors (plane dtypes only).
    `wino_always`: YV3_OPT_WINO_ALWAYS (``net.winograd = "always"``: the Winograd form on every eligible layer, whatever the
    tile count)."""
    sp = pc.spec
    d = ConvDesc()
    d.options, d.big_tile_min = tuning_options()
    if two_lanes:
        d.options |= _ffi.OPT_TWO_LANES
    if wino_always:
        d.options |= _ffi.OPT_WINO_ALWAYS
    if wino4_tiles:
        d.options |= _ffi.OPT_WINO4_TILES
    for i, v in enumerate((measure_env("YV3_TUNE", "") or "0").split(",")[:4]):
        d.tune[i] = int(v or 0)
    d.x, d.x2, d.w = _ptr(x), _ptr(x2), _ptr(pc.w)
    d.alpha, d.beta = _ptr(pc.alpha), _ptr(pc.beta)
    d.residual, d.y = _ptr(residual), _ptr(y)
    d.B, d.H, d.W = B, H, W
    d.cin, d.cin_up, d.cout, d.cout_pad = sp.cin, cin_up, sp.cout, pc.cout_pad
    d.k, d.stride = sp.k, sp.stride
    d.act = ACT_LEAKY if sp.bn else ACT_LINEAR
    d.dtype = dtype
    d.out_dtype = dtype if out_dtype is None else out_dtype
    d.flags = _ptr(flags)
    d.workspace = _ptr(workspace)
    d.workspace_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    if wino_ws is not None and pc.w_wino is not None and batch is None and dtype in (F32H2, F32) and (out_dtype is None or out_dtype == dtype):
        d.w_wino, d.alpha_wino = _ptr(pc.w_wino), _ptr(pc.alpha_wino)
        d.wino_ws, d.wino_ws_bytes = _ptr(wino_ws), wino_ws.numel() * wino_ws.element_size()
        if dtype == F32 and pc.w_wino4 is not None:
            d.w_wino4 = _ptr(pc.w_wino4)
    if batch is not None:
        b0, nb = batch
        ho, wo = out_hw(H, W, sp.k, sp.stride)

        def sl(t, per_img):                               # pointer to image b0 of plane 0; plane stride stays the full tensor's
            return ctypes.c_void_p(t.data_ptr() + b0 * per_img * t.element_size()) if t is not None else None
        cx = cin_up if cin_up else sp.cin
        hx, wx = (H // 2, W // 2) if cin_up else (H, W)
        d.x, d.x_plane_stride = sl(x, hx * wx * cx), B * hx * wx * cx
        if x2 is not None:
            d.x2, d.x2_plane_stride = sl(x2, H * W * (sp.cin - cin_up)), B * H * W * (sp.cin - cin_up)
        d.y, d.y_plane_stride = sl(y, ho * wo * sp.cout), B * ho * wo * sp.cout
        d.residual = sl(residual, ho * wo * sp.cout)
        d.B = nb
    return d


def out_hw(h, w, k, stride):
    pad = (k - 1) // 2
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


class Plan:
    """Buffers + kernel descriptors for one (B, H, W)."""

    @staticmethod
    def geometry(engine, H, W):
        """(detection rows N per image, attributes 5+C) of a plan for H x W inputs, without allocating one."""
        if H % 32 or W % 32:
            raise _ffi.Yv3Error("input height/width must be multiples of 32 (got %dx%d)" % (H, W))
        return 3 * sum((H // s) * (W // s) for s in (32, 16, 8)), 5 + engine.num_class

    def __init__(self, engine, B, H, W, flags=None, two_lanes=False):
        """two_lanes: this plan is one of two equal lanes that run concurrently (`Detector`): tells the library's per-launch
        kernel choice that the chip is shared (yv3.h YV3_OPT_TWO_LANES)."""
        if H % 32 or W % 32:
            raise _ffi.Yv3Error("input height/width must be multiples of 32 (got %dx%d)" % (H, W))
        self.B, self.H, self.W = B, H, W
        dev, dt = engine.device, engine.dtype
        packed = engine.packed
        nc = engine.num_class
        attrib = 5 + nc
        keep = []            # every buffer the descriptors point to
        descs = []
        # sticky status word written by the kernels (bit 0: an fp16-plane output was saturated) + its host mirror
        self.flags = flags if flags is not None else torch.zeros(1, device=dev, dtype=torch.int32)   # (lanes of one Detector share one)
        self.flags_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.flags_event = None
        # scratch of the stream-K schedule (fp16-plane kernels): zero-filled once, one per plan (= per launch stream)
        # automatic: small one-lane batches -- up to SK_AUTO_CELLS 32x32-pixel cells in the batch (8 images of 416 x 416), where most
        # layers have fewer tiles than the chip has CUs and the even split over (tile, K chunk) fills them: same box, alternating
        # (profiles/r04ag_stream_k_small_batches_ab.txt) 416x416 bs=4 +17 %, bs=8 +4 %; bs=16 -0.8 %, bs=32 -2 %, 608x608 bs=8 / 16 -4 %
        cells = B * ((H + 31) // 32) * ((W + 31) // 32)
        use_sk = engine.stream_k if engine.stream_k is not None else (cells <= SK_AUTO_CELLS and not two_lanes)
        self.workspace = (torch.zeros(_ffi.lib().yv3_conv_workspace_bytes(), device=dev, dtype=torch.uint8)
                          if (dt == F32H2 and use_sk) else None)
        self.layer_out = {}  # conv name -> (buffer, (h, w, c)) for bring-up / per-layer parity tests
        # Winograd scratch (the transformed input of ONE layer at a time; launches of a plan are stream-ordered): sized for
        # the largest eligible layer of this plan
        self.wino_ws = None
        if engine.winograd and dt in (F32H2, F32) and not engine.batch_split:      # (batch_split slices the same layers: direct kernels only)
            wsb = _ffi.lib().yv3_wino_workspace_bytes
            # eligible layers read 64 channels at H/4 (F32 only), 128 at H/8, 256 at H/16, 512 at H/32: the shallowest one is the largest
            lo = WINO_MIN_CIN_F32 if dt == F32 else WINO_MIN_CIN
            need = max(wsb(B, H // f, W // f, c) for c, f in ((64, 4), (128, 8), (256, 16), (512, 32)) if c >= min(lo, 512))
            if dt == F32:          # (the F(4x4) form's V is 36 positions x a sixteenth of the pixels: never the larger one, asked for the record)
                need = max(need, max(_ffi.lib().yv3_wino4_workspace_bytes(B, H // f, W // f, c) for c, f in ((64, 4), (128, 8), (256, 16), (512, 32))))
            self.wino_ws = torch.zeros(need, device=dev, dtype=torch.uint8)       # (zero-filled: hand-over flags of the even schedule)

        def buf(h, w, c, dtype=dt):
            t = alloc_act(B, h, w, c, dtype, dev)
            keep.append(t)
            return t

        # the YOLO decode is fused into the three head convs' epilogue (plane kernels): their logits are then not
        # materialised and the descriptors' dec_out is pointed at the caller's detections tensor before each launch
        self.fused_decode = bool(engine.fuse_decode and dt != F32)
        self.head_descs = []     # (descriptor index, ho, wo)
        self.desc_spec = []      # descriptor index -> index into arch.conv_specs (a layer may run as two launches: batch_split)
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        ncu = ncu & ~7 if ncu >= 8 else 256
        # feature.mlist.0 + feature.mlist.1 run as ONE launch (csrc/conv_front.hip; exact fp32: csrc/conv_front_f32.hip); the first
        # layer's [B,H,W,32] activation is then never written (conv0_out stays allocated for the un-fused / layer-by-layer paths)
        self.fused_front = bool(engine.fuse_front and dt in (F32H2, BF16, F32))
        # ... and the first residual block (feature.mlist.2: 1x1 64->32 + 3x3 32->64 + add) as one more (csrc/conv_res64.hip,
        # csrc/conv_res64_f32.hip)
        self.fused_res64 = bool(self.fused_front and engine.fuse_res64)
        self.first_desc = 3 if self.fused_res64 else (1 if self.fused_front else 0)

        def conv(i, x, h, w, residual=None, x2=None, cin_up=0, out_dtype=None):
            pc = packed[i]
            ho, wo = out_hw(h, w, pc.spec.k, pc.spec.stride)
            head = out_dtype == F32 and dt != F32
            y = None if (head and self.fused_decode) else buf(ho, wo, pc.spec.cout, dt if out_dtype is None else out_dtype)
            if head:
                self.head_descs.append((len(descs), ho, wo))
            split = None
            if engine.batch_split and dt == F32H2 and pc.spec.k == 3 and not head and self.workspace is None:
                split = batch_split(B, ho, wo, pc.cout_pad, ncu)
            for part in ([None] if split is None else [(0, split[0]), (split[0], split[1])]):
                descs.append(make_desc(pc, x, y, B, h, w, residual, x2, cin_up, dt, out_dtype, self.flags, self.workspace, batch=part,
                                       wino_ws=self.wino_ws, two_lanes=two_lanes, wino_always=engine.wino_always,
                                       wino4_tiles=engine.stream_k is False))
                self.desc_spec.append(i)
            if y is not None:
                self.layer_out[pc.spec.name] = y
            return y, ho, wo

        # ---- backbone (reference darknet.py:72-88)
        self.conv0_out = buf(H, W, 32)
        self.layer_out["feature.mlist.0"] = self.conv0_out
        cur, h, w = self.conv0_out, H, W
        i = 1
        routes = {}
        for stage, nblk in enumerate(arch.BACKBONE_BLOCKS):
            cur, h, w = conv(i, cur, h, w); i += 1
            for _ in range(nblk):
                mid, _, _ = conv(i, cur, h, w); i += 1
                cur, _, _ = conv(i, mid, h, w, residual=cur); i += 1          # x + conv2(conv1(x)), :53
            routes[stage] = (cur, h, w)
        r36, r61 = routes[2], routes[3]        # mlist[14] 52x52x256, mlist[23] 26x26x512 (:180-181)

        # ---- detection branches (darknet.py:204-223)
        self.logits = []

        def branch(x, h, w, x2=None, cin_up=0):
            nonlocal i
            route = None
            for j in range(6):
                if j == 0 and x2 is not None:
                    x, _, _ = conv(i, x, h, w, x2=x2, cin_up=cin_up)
                else:
                    x, _, _ = conv(i, x, h, w)
                i += 1
                if j == 4:
                    route = x                                                   # addCachedOut(-3), :185
            lg, _, _ = conv(i, x, h, w, out_dtype=F32); i += 1
            self.logits.append((lg, h, w))
            return route

        h1 = branch(cur, h, w)
        u1, _, _ = conv(i, h1, h, w); i += 1                                      # up1.conv (13x13)
        t61, h61, w61 = r61
        h2 = branch(u1, h61, w61, x2=t61, cin_up=packed[i - 1].spec.cout)         # nearest x2 + cat fused
        u2, _, _ = conv(i, h2, h61, w61); i += 1                                  # up2.conv (26x26)
        t36, h36, w36 = r36
        branch(u2, h36, w36, x2=t36, cin_up=packed[i - 1].spec.cout)
        assert i == len(packed) == 75 and self.desc_spec[:3] == [1, 2, 3]

        self._keep = keep
        self.n_desc = len(descs)
        self.descs = (ConvDesc * len(descs))(*descs)
        self.rows = [hh * ww * 3 for (_, hh, ww) in self.logits]
        self.N = sum(self.rows)
        self.attrib = attrib
        self._dets_ptr = None
        # decode parameters (yololayer.py:36-38): stride = H_img / nH, anchors by mask
        anchors = engine.anchors
        self.decode_args = []
        row0 = 0
        for (lg, hh, ww), mask in zip(self.logits, arch.ANCHOR_MASKS):
            flat = []
            for m in mask:
                flat += [float(anchors[2 * m]), float(anchors[2 * m + 1])]
            self.decode_args.append(((ctypes.c_float * 6)(*flat), float(H) / hh, row0, lg, hh, ww))
            row0 += hh * ww * 3
        if self.fused_decode:
            for (di, ho, wo), (anc, stride, r0, _, _, _) in zip(self.head_descs, self.decode_args):
                d = self.descs[di]
                d.dec_stride = stride
                for k in range(6):
                    d.dec_anchors[k] = anc[k]
                d.dec_out_batch_stride = self.N * attrib

    def bind_detections(self, dets):
        """Point the fused head convs at `dets` ([B, N, 5+C] fp32, contiguous)."""
        if not self.fused_decode or self._dets_ptr == dets.data_ptr():
            return
        for (di, _, _), (_, _, r0, _, _, _) in zip(self.head_descs, self.decode_args):
            self.descs[di].dec_out = dets.data_ptr() + r0 * self.attrib * 4
        self._dets_ptr = dets.data_ptr()

    def bytes_allocated(self):
        return sum(t.numel() * t.element_size() for t in self._keep)

    def launches(self):
        """Kernel launches per descriptor of the launch sequence (``yv3_conv2d_launches``): 2 for the Winograd forms and for a 1x1 layer that
        runs as persistent GEMM + small tiles; profiling tools map a kernel trace to layers with it."""
        lib = _ffi.lib()
        out = []
        for j in range(self.first_desc, self.n_desc):
            n = lib.yv3_conv2d_launches(ctypes.byref(self.descs[j]))
            if n < 0:
                _ffi.check(n, "yv3_conv2d_launches")
            out.append(int(n))
        return out

    def forms(self):
        """Per descriptor of the launch sequence (``descs[first_desc:]``): (conv spec index, form) with form = 0 direct /
        1 Winograd F(2x2,3x3) / 2 Winograd F(4x4,3x3) (exact-fp32 mode), as the library decides it on the current device
        (``yv3_conv2d_form``: tile count, lane count, CU count)."""
        lib = _ffi.lib()
        out = []
        for j in range(self.first_desc, self.n_desc):
            f = lib.yv3_conv2d_form(ctypes.byref(self.descs[j]))
            if f < 0:
                _ffi.check(f, "yv3_conv2d_form")
            out.append((self.desc_spec[j], int(f)))
        return out

    def executed_mac_factor(self):
        """{conv spec index: matrix multiplications executed / direct-form multiplications} for the launch sequence:
        16/36 for a launch that takes the Winograd F(2x2,3x3) form (even pictures; the tile grid of an odd picture is
        ceil(H/2) x ceil(W/2), so e.g. 13x13 executes 16 * 49 per 9 * 169 direct), 36/144 for F(4x4,3x3) (13x13: 36 * 16 per
        9 * 169), 1 otherwise."""
        out = {}
        for j, (si, f) in zip(range(self.first_desc, self.n_desc), self.forms()):
            d = self.descs[j]
            if f == 1:
                out[si] = (16.0 * ((d.H + 1) // 2) * ((d.W + 1) // 2)) / (9.0 * d.H * d.W)
            elif f == 2:                # F(4x4,3x3): 36 per 4x4 tile instead of 144 (tile grid ceil(H/4) x ceil(W/4))
                out[si] = (36.0 * ((d.H + 3) // 4) * ((d.W + 3) // 4)) / (9.0 * d.H * d.W)
            else:
                out[si] = 1.0
        return out


class Engine:
    """Packs a YoloNet's parameters and runs the static plan."""

    def __init__(self, net, dtype=F32):
        self.net = net
        self.dtype = dtype
        self.num_class = net.numClass
        self.anchors = [float(a) for a in net.anchors_flat]
        self.specs = arch.conv_specs(self.num_class)
        self.packed = None
        self.device = None
        self._sig = None
        self._plans = {}
        self.lane_choices = {}        # Detector: (shape, eval) -> (lanes, stream pair, calibration ms), measured once per engine
        self.generation = 0
        # stream-K schedule of the 13x13 layers (+1.7 % at 416x416 bs=64): opt-in, because a tile split between two
        # workgroups is summed in a batch-position-dependent order (include/yv3.h, yv3_conv_desc.workspace)
        # None (default): small one-lane batches only (SK_AUTO_CELLS: up to 8 images of 416 x 416 -- every layer has fewer tiles than the
        # chip has CUs, splitting their K ranges over the idle CUs cuts a single image's latency 2.25 -> 1.75 ms and gains 17 % at
        # bs=4); True / YV3_SK=1: wherever the library's shape rule applies; False / YV3_SK=0: never
        sk = getattr(net, "stream_k", None)
        if sk is None and measure_env("YV3_SK") in ("0", "1"):
            sk = measure_env("YV3_SK") == "1"
        self.stream_k = None if sk is None else bool(sk)
        self.fuse_decode = bool(getattr(net, "fuse_decode", measure_env("YV3_NO_FUSED_DECODE") is None))
        self.fuse_front = bool(getattr(net, "fuse_front", measure_env("YV3_NO_FUSED_FRONT") is None))
        self.fuse_res64 = bool(getattr(net, "fuse_res64", measure_env("YV3_NO_FUSED_RES64") is None))
        # opt-in (measured null end to end, see batch_split): 13x13 layers as "one full round" + "the rest"
        self.batch_split = bool(getattr(net, "batch_split", measure_env("YV3_BATCH_SPLIT") == "1"))
        # Winograd F(2x2,3x3) for the 3x3 stride-1 layers with >= 256 input channels (fp16-plane mode; csrc/winograd.hip)
        # (default on: the library uses it per launch when the tile count suits it, yv3_conv_desc.w_wino; YV3_WINO=0 / net.winograd = False: never)
        # net.winograd = "always" (YV3_WINO=always): on every eligible layer whatever the tile count (parity tests of the form itself)
        wino = getattr(net, "winograd", None)
        if wino is None:
            env = measure_env("YV3_WINO", "1")
            wino = "always" if env == "always" else env != "0"
        self.wino_always = wino == "always" or bool(measure_env("YV3_WINO_ALWAYS"))
        self.winograd = bool(wino)
        # exact-fp32 mode: F(4x4,3x3) (csrc/conv_wino4_f32.hip) wherever the library's rule takes it; net.winograd4 = False keeps F(2x2,3x3)
        self.winograd4 = bool(getattr(net, "winograd4", measure_env("YV3_WINO4", "1") != "0"))
        # net.deterministic = True: ONE switch for "the same image gives the same bits at every batch size, batch position and lane
        # count": direct one-tile-per-workgroup kernels only (no per-launch Winograd choice, no stream-K split tiles); `Detector`
        # then also runs a single lane.  Costs ~10 % at bs=64.
        self.deterministic = bool(getattr(net, "deterministic", measure_env("YV3_DETERMINISTIC") == "1"))
        if self.deterministic:
            self.winograd, self.wino_always, self.stream_k = False, False, False

    # -- weights
    def _param_tensors(self):
        """Every tensor the packed weights depend on, in spec order.  The module objects are resolved once (75
        ``get_submodule`` path walks cost ~1.4 ms -- per forward, and exposed in every synchronous ``detect()`` call); the
        tensors themselves are re-read from them each time, so re-assigned parameters are still seen."""
        slots = self.__dict__.get("_mod_slots")
        if slots is not None:
            # a cached module is only trusted while EVERY container on its path from the root still holds the same child under the
            # same name: the distinct (registry, key, child) links of all 75 paths (~110 dict look-ups, ~15 us) are re-checked, so a
            # replaced leaf (net.pre_det1.mlist[6] = nn.Conv2d(...)), a replaced container (net.pre_det1 = PreDetectionConvGroup(...))
            # or a replaced ModuleList (net.pre_det1.mlist = nn.ModuleList(...)) is re-resolved -- and, its tensors being new
            # objects, changes the signature (the leaf's parent alone is not enough)
            for reg, key, m in self._mod_links:
                if reg.get(key) is not m:
                    slots = None
                    break
        if slots is None:
            slots, links, seen = [], [], set()
            for sp in self.specs:
                node = self.net
                for part in sp.name.split("."):
                    child = node._modules[part]
                    if (id(node._modules), part) not in seen:
                        seen.add((id(node._modules), part))
                        links.append((node._modules, part, child))
                    node = child
                slots.append((None, None, node))
            self._mod_slots, self._mod_links = slots, links
        out = []
        for _, _, m in slots:                        # (straight from the modules' dicts: nn.Module.__getattr__ is slow)
            if isinstance(m, torch.nn.Conv2d):
                pr = m._parameters
                out.append(pr["weight"])
                if pr["bias"] is not None:
                    out.append(pr["bias"])
            else:
                sub = m._modules
                bn = sub["bn"]
                bp, bb = bn._parameters, bn._buffers
                out += [sub["conv"]._parameters["weight"], bp["weight"], bp["bias"], bb["running_mean"], bb["running_var"]]
        return out

    def _signature(self):
        """What the packed weights were built from.  (data_ptr, _version) of every parameter catches assignments,
        ``load_state_dict``, ``WeightManager`` loads and in-place ops on the parameter itself.  It does NOT see
        writes through ``param.data`` (``p.data.copy_()`` bumps the version counter of a temporary alias, not of
        ``p`` -- the reference's own loader idiom, darknet.py:275): after such edits call ``net.repack()``, or set
        ``net.weight_check = "checksum"`` to add a device-side position-sensitive hash of all parameters (one pass over
        the 248 MB of weights and one host sync per forward) to the signature."""
        ts = self._param_tensors()
        sig = tuple((t.data_ptr(), t._version) for t in ts)
        if getattr(self.net, "weight_check", "version") == "checksum" and ts and ts[0].is_cuda:
            # position-sensitive integer hash of the parameters' BIT patterns: sum_i bits_i * (i mod 65521 + 1) in wrapping
            # int64 per tensor, tensors combined with their index -- sees sign flips, swapped filters and swapped tensors
            # (an L1 norm does not).  Costs one pass over the 248 MB of weights AND one host sync per forward (it defeats
            # ``net.async_forward``): a debugging aid, not a default.
            with torch.no_grad():
                parts = []
                for t in ts:
                    v = t.detach().contiguous().view(-1)
                    v = (v.view(torch.int32) if v.dtype == torch.float32 else v.float().view(torch.int32)).to(torch.int64)
                    w = self._iota(v.numel(), v.device)
                    parts.append((v * w).sum())
                tot = torch.stack(parts) * torch.arange(1, len(parts) + 1, device=parts[0].device, dtype=torch.int64)
                chk = int(tot.sum().item())
            sig += (chk,)
        return sig

    def _iota(self, n, device):
        c = self.__dict__.setdefault("_iota_cache", {})
        w = c.get((n, device))
        if w is None:
            w = c[(n, device)] = torch.arange(n, device=device, dtype=torch.int64) % 65521 + 1
        return w

    def ensure_packed(self):
        sig = self._signature()
        if self.packed is not None and sig == self._sig:
            return
        first = self.net.get_submodule(self.specs[0].name).conv.weight
        _ffi.require_cuda(first, "YoloNet parameters (call net.cuda() first)")
        self.device = first.device
        with torch.cuda.device(self.device):
            self.packed = [pack_conv(self.net.get_submodule(sp.name), sp, self.dtype, self.winograd, self.winograd4) for sp in self.specs]
        self._sig = sig
        self._plans = {}
        self.generation += 1          # holders of a Plan (Detector) must rebuild: descriptors point into `packed`

    def invalidate(self):
        """Forget the packed weights (the next `ensure_packed` re-packs from the current parameters and bumps
        `generation`, so every holder of this engine -- a `Detector` the caller kept -- rebuilds its plans)."""
        self.packed = None
        self._sig = None
        self._plans = {}
        self.__dict__.pop("_mod_slots", None)

    # -- plans
    def drop_plan(self, B, H, W):
        self._plans.pop((B, H, W), None)

    def plan(self, B, H, W):
        key = (B, H, W)
        p = self._plans.get(key)
        if p is None:
            p = self._plans[key] = Plan(self, B, H, W)
        return p

    # -- execution
    def run_conv0(self, plan, x):
        """feature.mlist.0 (reads the caller's NCHW batch)."""
        p0 = self.packed[0]
        _ffi.check(_ffi.lib().yv3_conv0(x.data_ptr(), p0.w.data_ptr(), p0.alpha.data_ptr(), p0.beta.data_ptr(),
                                        plan.conv0_out.data_ptr(), plan.B, plan.H, plan.W, self.dtype, plan.flags.data_ptr(),
                                        _ffi.stream_ptr()), "yv3_conv0")

    def run_front(self, plan, x):
        """The network's front: feature.mlist.0 alone, or (fused-front plans) feature.mlist.0 + feature.mlist.1 in one launch
        whose output is descriptor 0's output buffer -- followed, in fused-res64 plans, by the first residual block
        (descriptors 1 and 2) as one launch."""
        if not plan.fused_front:
            return self.run_conv0(plan, x)
        p0, p1, d1 = self.packed[0], self.packed[1], plan.descs[0]
        if self.dtype == F32:
            _ffi.check(_ffi.lib().yv3_conv_front_f32(x.data_ptr(), p0.w.data_ptr(), p0.alpha.data_ptr(), p0.beta.data_ptr(),
                                                     p1.w.data_ptr(), p1.alpha.data_ptr(), p1.beta.data_ptr(), d1.y,
                                                     plan.B, plan.H, plan.W, _ffi.stream_ptr()), "yv3_conv_front_f32")
            if plan.fused_res64:
                p2, p3, d3 = self.packed[2], self.packed[3], plan.descs[2]
                _ffi.check(_ffi.lib().yv3_res_block64_f32(d1.y, p2.w.data_ptr(), p2.alpha.data_ptr(), p2.beta.data_ptr(),
                                                          p3.w.data_ptr(), p3.alpha.data_ptr(), p3.beta.data_ptr(), d3.y,
                                                          plan.B, plan.H // 2, plan.W // 2, _ffi.stream_ptr()), "yv3_res_block64_f32")
            return
        front = _ffi.lib().yv3_conv_front if self.dtype == F32H2 else _ffi.lib().yv3_conv_front_bf16
        _ffi.check(front(x.data_ptr(), p0.w.data_ptr(), p0.alpha.data_ptr(), p0.beta.data_ptr(),
                                             p1.w.data_ptr(), p1.alpha.data_ptr(), p1.beta.data_ptr(), d1.y,
                                             plan.B, plan.H, plan.W, plan.flags.data_ptr(), _ffi.stream_ptr()), "yv3_conv_front")
        if plan.fused_res64:
            p2, p3, d3 = self.packed[2], self.packed[3], plan.descs[2]
            res = _ffi.lib().yv3_res_block64 if self.dtype == F32H2 else _ffi.lib().yv3_res_block64_bf16
            _ffi.check(res(d1.y, p2.w.data_ptr(), p2.alpha.data_ptr(), p2.beta.data_ptr(),
                                                  p3.w.data_ptr(), p3.alpha.data_ptr(), p3.beta.data_ptr(), d3.y,
                                                  plan.B, plan.H // 2, plan.W // 2, plan.flags.data_ptr(), _ffi.stream_ptr()),
                       "yv3_res_block64")

    def run_conv_sequence(self, plan, dets=None):
        """The remaining convolutions (74, or 73 behind a fused front).  With a fused-decode plan `dets` (the detections
        tensor the head convs write) is required and `run_decode` is a no-op."""
        if plan.fused_decode:
            if dets is None:
                raise _ffi.Yv3Error("this plan decodes inside the head convs: pass the detections tensor")
            plan.bind_detections(dets)
        first = plan.first_desc
        tail = ctypes.cast(ctypes.addressof(plan.descs) + first * ctypes.sizeof(ConvDesc), ctypes.POINTER(ConvDesc))
        _ffi.check(_ffi.lib().yv3_conv2d_sequence(tail, plan.n_desc - first, _ffi.stream_ptr()), "yv3_conv2d_sequence")

    def run_convs(self, plan, x, dets=None):
        """All 75 convolutions."""
        self.run_front(plan, x)
        self.run_conv_sequence(plan, dets)

    def run_decode(self, plan, dets):
        if plan.fused_decode:
            return
        lib = _ffi.lib()
        s = _ffi.stream_ptr()
        bstride = plan.N * plan.attrib
        for anc, stride, row0, lg, hh, ww in plan.decode_args:
            out_ptr = dets.data_ptr() + row0 * plan.attrib * 4
            _ffi.check(lib.yv3_decode(lg.data_ptr(), lg.shape[-1], anc, stride, out_ptr, bstride,
                                      plan.B, hh, ww, self.num_class, s), "yv3_decode")

    def prepare_input(self, x):
        _ffi.require_cuda(x, "input images")
        if x.dim() != 4 or x.shape[1] != 3:
            raise _ffi.Yv3Error("expected images [B,3,H,W], got %s" % (tuple(x.shape),))
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        return x

    OVERFLOW_MSG = ("an activation exceeded the fp16 range (|v| > 65504) and was saturated: math mode F32H2 cannot "
                    "represent this network/input; set net.math_mode = yolo_v3_amd.F32X3 (or F32) and rerun "
                    "(net.strict_range = False, the default, does that by itself)")
    OVERFLOW_MSG_BF16 = ("the first layer of math mode BF16 runs on the fp16 matrix cores and needs |input| <= 4094 and "
                         "|weight| <= 255 (csrc/conv0.hip); this input / these weights exceed that and were saturated: "
                         "use net.math_mode = yolo_v3_amd.F32X3 (or F32)")

    def raise_if_overflowed(self, plan, flag_value):
        """Turn the kernels' sticky saturation flag into an error (and clear it)."""
        if flag_value:
            plan.flags.zero_()
            plan.flags_host.zero_()
            plan.flags_event = None
            if flag_value & 2:
                raise StreamKTimeout("a stream-K accumulator hand-over timed out: the schedule needs all of its workgroups resident at once "
                                     "and something else (another stream / thread / process running small batches) held part of the GPU.  "
                                     "Callers that share the GPU set net.stream_k = False (or net.deterministic = True)")
            raise RangeOverflow(self.OVERFLOW_MSG_BF16 if self.dtype == BF16 else self.OVERFLOW_MSG)

    def range_fallback(self):
        """After a RangeOverflow in the default mode: the engine that takes over -- F32X3, the same plane pipeline on three bf16
        planes (fp32's exponent range, 24 significant bits) -- or None when there is none to fall back to (``net.strict_range = True``:
        the caller wants the error; BF16: a reduced-precision mode the caller chose).  The choice is remembered on the network
        (``YoloNet.engine`` hands out the fall-back engine wherever F32H2 is asked for from now on) and announced once."""
        if self.dtype != F32H2 or getattr(self.net, "strict_range", False):
            return None
        import warnings
        if getattr(self.net, "_range_fallback", None) is None:
            warnings.warn("yolo_v3_amd: an activation left the fp16 range of math mode F32H2 (|v| > 65504); this network now runs in "
                          "F32X3 (three bf16 planes: fp32 range, 24 bits, ~1.9x the time) -- set net.math_mode = F32X3 / F32 to avoid the "
                          "first, discarded pass, or net.strict_range = True to get an error instead", RuntimeWarning)
            self.net._range_fallback = F32X3
        return self.net.engine(F32X3)

    def checks_status(self):
        """Does a forward of this engine have a kernel status word to look at?  F32H2 / BF16: fp16 saturation (+ stream-K hand-over);
        F32: only the F(4x4) stage's even schedule can set it (hand-over time-out on a shared GPU) -- not with ``net.stream_k = False``."""
        return self.dtype in (F32H2, BF16) or (self.dtype == F32 and self.winograd and self.winograd4 and self.stream_k is not False)

    def disable_stream_k(self):
        """After a StreamKTimeout: this engine stops using the stream-K schedule (plans are rebuilt without its workspace; holders
        of a Plan -- `Detector` -- see the new `generation`).  The results of the call that timed out are invalid and must be
        recomputed by the caller (`forward`, `Detector.__call__` and `detect` do)."""
        import warnings
        warnings.warn("yolo_v3_amd: stream-K hand-over timed out (the GPU is shared with another small-batch caller); the stream-K "
                      "schedule is now off for this network -- set net.stream_k = False to avoid the first slow call", RuntimeWarning)
        self.stream_k = False
        self._plans = {}
        self.generation += 1

    def forward(self, x, dets=None, _retry=True):
        """x: [B,3,H,W] fp32 on the GPU -> detections [B, N, 5+C] (cx,cy,w,h,conf,cls...).

        In F32H2 mode (and BF16, whose first layer splits its operands into fp16) the kernels' saturation flag of THIS call is copied to pinned host memory behind the last
        kernel and checked before returning (one event wait: the call then returns with the work complete, like the
        reference's forward followed by any use of its result), so no entry point built on it -- ``net(x)``,
        ``forward_cat``, ``detect(is_eval=True)``, ``predict_and_process`` -- can hand out saturated values.
        ``net.async_forward = True`` restores the fully asynchronous behaviour (the flag of a call is then examined
        at the start of the next call on the same plan); `Detector` checks the flag with its single D2H copy of the
        box counts either way.  F32X3 has fp32's exponent range and never waits; exact F32 waits the same way while its F(4x4,3x3) stage may use the
        even schedule (``checks_status``: a hand-over time-out on a shared GPU is the one status it can report; not under ``net.stream_k = False``)."""
        x = self.prepare_input(x)
        with torch.cuda.device(x.device):
            self.ensure_packed()
            B, _, H, W = x.shape
            plan = self.plan(B, H, W)
            if self.checks_status() and plan.flags_event is not None and plan.flags_event.query():
                try:
                    self.raise_if_overflowed(plan, int(plan.flags_host[0]))
                except StreamKTimeout:                    # (net.async_forward: the EARLIER call's result is invalid -- say so -- but repair the future)
                    self.disable_stream_k()
                    raise
                except RangeOverflow:
                    self.range_fallback()
                    raise
            if dets is None:
                dets = torch.empty((B, plan.N, plan.attrib), device=x.device, dtype=torch.float32)
            self.run_convs(plan, x, dets)
            self.run_decode(plan, dets)
            if self.checks_status():
                plan.flags_host.copy_(plan.flags, non_blocking=True)
                plan.flags_event = torch.cuda.Event()
                plan.flags_event.record()
                if not getattr(self.net, "async_forward", False):
                    plan.flags_event.synchronize()
                    try:
                        self.raise_if_overflowed(plan, int(plan.flags_host[0]))
                    except StreamKTimeout:
                        if not _retry:
                            raise
                        self.disable_stream_k()               # (fall back instead of failing the call)
                        return self.forward(x, dets, _retry=False)
                    except RangeOverflow:                     # (the default mode computes where the reference computes)
                        fb = self.range_fallback()
                        if fb is None:
                            raise
                        return fb.forward(x, dets)
                else:
                    # asynchronous callers: a time-out / saturation seen at the start of the NEXT call cannot be repaired for the call that
                    # produced it; the schedule / mode is switched for the calls that follow and the error still tells the caller that the
                    # earlier result is invalid (raise_if_overflowed at the top of this function)
                    pass
        return dets, plan
