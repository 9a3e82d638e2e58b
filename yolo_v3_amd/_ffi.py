"""ctypes binding of libyv3.so (C-ABI declared in include/yv3.h).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C yolo_v3_amd/csrc``).
There is deliberately no fallback: if the shared object is missing or a call fails, the
product path raises.  PyTorch is only used by callers for device memory and streams.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YV3_LIB (another build of the library: A/B variants, the -DYV3_MEASURE build) is honoured in measurement sessions only (YV3_MEASURE=1)
LIB_PATH = (os.environ.get("YV3_LIB") if os.environ.get("YV3_MEASURE") == "1" else None) or os.path.join(_HERE, "libyv3.so")

if os.environ.get("YV3_MEASURE") != "1":
    _stray = sorted(k for k in os.environ if k.startswith("YV3_") and k not in ("YV3_MEASURE", "YV3_DIST_BACKEND", "YV3_DUMP_PLAN"))
    if _stray:                      # (a user who points YV3_LIB at another build must learn that it is not loaded)
        import warnings
        warnings.warn("yolo_v3_amd: %s set but ignored -- library / kernel-selection overrides are honoured in measurement sessions only "
                      "(YV3_MEASURE=1); the bundled libyv3.so and the default plans are used" % ", ".join(_stray), RuntimeWarning)

F32, BF16, F32X3, F32H2 = 0, 1, 2, 3
ACT_LINEAR, ACT_LEAKY = 0, 1
PP_EVAL, PP_PROB = 1, 2
OPT_NO_PINGPONG, OPT_K3S1, OPT_WINO_EVEN, OPT_WINO_ALWAYS, OPT_TWO_LANES, OPT_WINO4_TILES = 1, 2, 4, 8, 16, 32

c_void_p, c_int, c_float, c_size_t, c_longlong = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                  ctypes.c_size_t, ctypes.c_longlong)


class ConvDesc(ctypes.Structure):
    """struct yv3_conv_desc (include/yv3.h)."""
    _fields_ = [("x", c_void_p), ("x2", c_void_p), ("w", c_void_p), ("alpha", c_void_p), ("beta", c_void_p),
                ("residual", c_void_p), ("y", c_void_p),
                ("B", c_int), ("H", c_int), ("W", c_int), ("cin", c_int), ("cin_up", c_int),
                ("cout", c_int), ("cout_pad", c_int), ("k", c_int), ("stride", c_int), ("act", c_int),
                ("dtype", c_int), ("out_dtype", c_int), ("flags", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("dec_out", c_void_p), ("dec_out_batch_stride", c_longlong), ("dec_stride", c_float),
                ("dec_anchors", c_float * 6), ("options", ctypes.c_uint), ("big_tile_min", c_int), ("tune", c_int * 4),
                ("x_plane_stride", c_longlong), ("x2_plane_stride", c_longlong), ("y_plane_stride", c_longlong),
                ("w_wino", c_void_p), ("alpha_wino", c_void_p), ("wino_ws", c_void_p), ("wino_ws_bytes", ctypes.c_size_t),
                ("w_wino4", c_void_p)]


_SIGNATURES = {
    "yv3_version": (c_int, []),
    "yv3_conv_workspace_bytes": (ctypes.c_size_t, []),
    "yv3_wino_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    "yv3_wino4_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    "yv3_pack_wino4_weight_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "yv3_error_string": (ctypes.c_char_p, [c_int]),
    "yv3_pack_conv_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "yv3_fold_bn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "yv3_split_planes": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_void_p]),
    "yv3_merge_planes": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_void_p]),
    "yv3_conv0": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "yv3_conv_front": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "yv3_conv_front_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "yv3_res_block64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "yv3_res_block64_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "yv3_conv_front_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "yv3_res_block64_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "yv3_conv2d": (c_int, [ctypes.POINTER(ConvDesc), c_void_p]),
    "yv3_conv2d_form": (c_int, [ctypes.POINTER(ConvDesc)]),
    "yv3_conv2d_launches": (c_int, [ctypes.POINTER(ConvDesc)]),
    "yv3_conv2d_sequence": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_void_p]),
    "yv3_decode": (c_int, [c_void_p, c_int, ctypes.POINTER(c_float), c_float, c_void_p, c_longlong,
                           c_int, c_int, c_int, c_int, c_void_p]),
    "yv3_decode_nchw": (c_int, [c_void_p, ctypes.POINTER(c_float), c_float, c_void_p, c_longlong,
                                c_int, c_int, c_int, c_int, c_void_p]),
    "yv3_cxcywh_to_xyxy": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p]),
    "yv3_iou_matrix": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "yv3_letterbox": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "yv3_letterbox_ex": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "yv3_resize_linear": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "yv3_upsample2x_concat": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "yv3_correct_boxes": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "yv3_gather_boxes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "yv3_postproc_cand_bytes": (c_size_t, [c_int, c_int, c_int]),
    "yv3_postproc_nms_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "yv3_postproc_filter": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "yv3_postproc_nms": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int, c_void_p, c_int,
                                 c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None


class Yv3Error(RuntimeError):
    pass


def lib():
    """Load libyv3.so once; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Yv3Error("HIP extension %s is missing: build it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C yolo_v3_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().yv3_error_string(rc).decode()
        raise Yv3Error("%s failed: %s (code %d)" % (what or "libyv3 call", msg, rc))


def stream_ptr():
    """hipStream_t of torch's current stream on the current device."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(t, name="tensor"):
    if not t.is_cuda:
        raise Yv3Error("%s must live on the GPU: this package runs only on MI355X (HIP kernels), "
                       "there is no CPU path" % name)
    return t
