"""Drop-in surface of the hot-path part of the reference's ``utils.py``.

``postprocessing`` (reference utils.py:226-258) with ``get_nms_detections`` (utils.py:148-202) /
``get_raw_detections`` (utils.py:204-224), ``iou_vectorized`` (utils.py:98-119) and ``bbox_iou``
(utils.py:122-146) -- same signatures and result conventions, computed by HIP kernels
(``csrc/postproc.hip``) instead of a CPU Python loop:

* result is ``[]`` when no image has a candidate (utils.py:248,251), otherwise a list of B CPU
  tensors ``[n_i, 7] = x1,y1,x2,y2,conf,score,cls``; an image without candidates yields
  ``torch.Tensor()`` (shape ``(0,)``, utils.py:153-158);
* per image: classes ascending, within a class score descending (ties: lower row first),
  strict ``>`` for both thresholds, zero-area boxes dropped (NaN self-IOU, utils.py:182);
* unlike the reference (utils.py:227-233) the caller's tensor is never modified.
"""
from collections import OrderedDict

import torch

from . import _ffi


def _as_gpu_f32(t, name):
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise _ffi.Yv3Error("%s is a CPU tensor and no GPU is available: this package has no CPU path" % name)
        t = t.cuda()
    return t.detach().float().contiguous()


class PostProcessor:
    """Reusable buffers for filter + NMS on ``[B, N, 5+C]`` detections (one per shape/device).

    ``counts`` / ``out``: optional views into buffers owned by the caller -- ``counts = (cand [B], kept [B])`` int32,
    ``out`` ``[B, cap, 7]`` fp32 -- so that several processors (the lanes of a `Detector`, each on its own stream)
    fill disjoint image ranges of ONE result tensor and ONE counts buffer (a single D2H copy fetches all of it)."""

    def __init__(self, B, N, num_classes, device, max_cand=None, cap=None, counts=None, out=None):
        self.B, self.N, self.C = B, N, num_classes
        self.device = torch.device(device)
        lib = _ffi.lib()
        self.max_cand = int(max_cand or N)
        self.cap = int(cap or self.max_cand)
        self.cand = torch.empty(lib.yv3_postproc_cand_bytes(B, self.max_cand, num_classes), dtype=torch.uint8, device=device)
        if counts is None:
            # [0:B] candidate counts, [B:2B] kept counts -- one buffer so a single D2H copy fetches both
            self.counts = torch.zeros(2 * B, dtype=torch.int32, device=device)
            self.cand_counts, self.kept_counts = self.counts[:B], self.counts[B:]
        else:
            self.counts = None
            self.cand_counts, self.kept_counts = counts
            assert self.cand_counts.numel() == B and self.kept_counts.numel() == B
        self._ws = None
        self._ws_n = 0
        self._out = out
        self._owns_out = out is None
        if out is not None:
            assert out.is_contiguous() and tuple(out.shape) == (B, self.cap, 7)

    def _workspace(self, max_n):
        lib = _ffi.lib()
        if self._ws is None or max_n > self._ws_n:
            self._ws = torch.empty(lib.yv3_postproc_nms_workspace_bytes(self.B, max_n, self.C), dtype=torch.uint8, device=self.device)
            self._ws_n = max_n
        return self._ws

    def bytes_allocated(self):
        """Device bytes this processor owns (candidate keys, NMS workspace, its own output / counts)."""
        own = [self.cand, self._ws, self.counts] + ([self._out] if self._owns_out else [])
        return sum(t.numel() * t.element_size() for t in own if t is not None)

    def filter(self, dets, conf_thr, is_eval, prob=False):
        mode = (_ffi.PP_EVAL if is_eval else 0) | (_ffi.PP_PROB if prob else 0)
        _ffi.check(_ffi.lib().yv3_postproc_filter(dets.data_ptr(), self.B, self.N, self.C, float(conf_thr), mode,
                                                  self.cand.data_ptr(), self.max_cand, self.cand_counts.data_ptr(),
                                                  _ffi.stream_ptr()), "yv3_postproc_filter")

    def nms(self, dets, nms_thr, use_nms, max_n, cap):
        ws = self._workspace(max_n)
        if self._out is None or self._out.shape[1] < cap:
            self._out = torch.empty((self.B, cap, 7), dtype=torch.float32, device=self.device)
        out = self._out
        _ffi.check(_ffi.lib().yv3_postproc_nms(dets.data_ptr(), self.B, self.N, self.C, float(nms_thr), int(bool(use_nms)),
                                               self.cand.data_ptr(), self.max_cand, self.cand_counts.data_ptr(), max_n,
                                               out.data_ptr(), out.shape[1], self.kept_counts.data_ptr(),
                                               ws.data_ptr(), ws.numel(), _ffi.stream_ptr()), "yv3_postproc_nms")
        return out

    def run_sync_free(self, dets, conf_thr, nms_thr, is_eval, use_nms, prob=False):
        """filter + NMS with worst-case buffers (max_cand candidates / cap boxes per image); nothing
        is read back.  Returns (boxes [B,cap,7], counts [2B]) on the GPU."""
        self.filter(dets, conf_thr, is_eval, prob)
        out = self.nms(dets, nms_thr, use_nms, self.max_cand, self.cap)
        return out, self.counts

    def to_list(self, out, counts_host):
        """Reference result convention from the device buffers + host copy of the counts."""
        return boxes_to_list(out, counts_host, self.B, self.max_cand)


def boxes_to_list(out, counts_host, B, max_cand):
    """``[B,cap,7]`` device boxes + host counts (``[0:B]`` candidates, ``[B:2B]`` kept) -> the reference's result
    convention (utils.py:148-202,248-251)."""
    ncand, nkeep = counts_host[:B].tolist(), counts_host[B:2 * B].tolist()
    if max(ncand) > max_cand:
        raise _ffi.Yv3Error("candidate buffer overflow (%d > %d): raise max_cand" % (max(ncand), max_cand))
    if max(nkeep) > out.shape[1]:
        raise _ffi.Yv3Error("box buffer overflow (%d > %d): raise cap" % (max(nkeep), out.shape[1]))
    if sum(ncand) == 0:
        return []                                        # utils.py:248,251
    kmax = max(nkeep)
    host = out[:, :max(kmax, 1)].cpu()
    return [host[b, :nkeep[b]].clone() if ncand[b] else torch.Tensor() for b in range(B)]


_PP_CACHE = OrderedDict()       # (device, stream, thread, B, N, C, max_cand) -> PostProcessor
_PP_CACHE_MAX = 4               # entries ...
_PP_CACHE_MAX_BYTES = 4 << 30   # ... and bytes (eval mode: 8*N*C B of keys per image + B*nmax*ceil(nmax/64)*8 B of NMS masks)


def clear_postproc_cache():
    """Drop the `postprocessing()` buffers kept for repeated calls (they can reach GBs in eval mode)."""
    _PP_CACHE.clear()


def _cached_postprocessor(B, N, num_classes, device, max_cand):
    """One PostProcessor per (shape, STREAM, THREAD): `postprocessing()` hands the same candidate / count / output / workspace
    buffers to every call with the same key, so calls that may overlap -- other streams, other threads -- must not share
    an entry.  Bounded by entry count and by bytes, least recently used first out."""
    import threading
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, threading.get_ident(), B, N, num_classes, max_cand)
    pp = _PP_CACHE.get(key)
    if pp is None:
        pp = PostProcessor(B, N, num_classes, device, max_cand=max_cand)
        _PP_CACHE[key] = pp
        while len(_PP_CACHE) > 1 and (len(_PP_CACHE) > _PP_CACHE_MAX or
                                      sum(q.bytes_allocated() for q in _PP_CACHE.values()) > _PP_CACHE_MAX_BYTES):
            _PP_CACHE.popitem(last=False)
    else:
        _PP_CACHE.move_to_end(key)
    return pp


def postprocessing(detections, num_classes, obj_conf_thr=0.5, nms_thr=0.4, is_eval=False, use_nms=True):
    """reference utils.py:226-258, on the GPU.  ``detections``: ``[B, N, 5+num_classes]``.

    Buffers are sized from the actual candidate counts (one extra host sync between filter and NMS: eval mode can
    produce up to N*C candidates per image); the processor -- in eval mode 8*N*C bytes of keys per image -- is
    cached per shape, so repeated calls allocate nothing."""
    det = _as_gpu_f32(detections, "detections")
    if det.dim() != 3 or det.shape[2] < 5 + num_classes:
        raise _ffi.Yv3Error("detections must be [B, N, >=5+num_classes]")
    if det.shape[2] != 5 + num_classes:
        det = det[..., :5 + num_classes].contiguous()
    B, N, _ = det.shape
    if B == 0 or N == 0:
        return []
    with torch.cuda.device(det.device):
        max_cand = N * num_classes if is_eval else N
        pp = _cached_postprocessor(B, N, num_classes, det.device, max_cand)
        pp.filter(det, obj_conf_thr, is_eval)
        ncand = pp.counts[:B].cpu()
        nmax = int(ncand.max())
        if nmax == 0:
            return []
        out = pp.nms(det, nms_thr, use_nms, nmax, nmax)
        return pp.to_list(out, pp.counts.cpu())


def iou_vectorized(bbox):
    """All-pairs IOU of x1y1x2y2 boxes ``[n, >=4]`` -> ``[n, n]`` (reference utils.py:98-119)."""
    return _iou(bbox, bbox, 0)


def bbox_iou(b1, b2, mode="x1y1x2y2"):
    """IOU matrix ``[n1, n2]`` for boxes in x1y1x2y2 or cxcywh form (reference utils.py:122-146)."""
    if mode not in ("x1y1x2y2", "cxcywh"):
        raise ValueError("mode must be 'x1y1x2y2' or 'cxcywh'")
    return _iou(b1, b2, 0 if mode == "x1y1x2y2" else 1)


def _iou(b1, b2, mode):
    was_cpu = not b1.is_cuda
    g1, g2 = _as_gpu_f32(b1, "boxes"), _as_gpu_f32(b2, "boxes")
    if g1.dim() != 2 or g2.dim() != 2 or g1.shape[1] < 4 or g2.shape[1] < 4:
        raise _ffi.Yv3Error("boxes must be [n, >=4]")
    n1, n2 = g1.shape[0], g2.shape[0]
    out = torch.empty((n1, n2), dtype=torch.float32, device=g1.device)
    if n1 and n2:
        with torch.cuda.device(g1.device):
            _ffi.check(_ffi.lib().yv3_iou_matrix(g1.data_ptr(), n1, g1.shape[1], g2.data_ptr(), n2, g2.shape[1], mode,
                                                 out.data_ptr(), _ffi.stream_ptr()), "yv3_iou_matrix")
    return out.cpu() if was_cpu else out


# ----------------------------------------------------------------------------- input preparation
def letterbox_transforms(inner_dim, outer_dim):
    """reference utils.py:34-42: (box_w, box_h, box_x_offset, box_y_offset, ratio) of an aspect-preserving fit of
    inner_dim = (w,h) into outer_dim = (w,h)."""
    outer_w, outer_h = outer_dim
    inner_w, inner_h = inner_dim
    ratio = min(outer_w / inner_w, outer_h / inner_h)
    box_w, box_h = int(inner_w * ratio), int(inner_h * ratio)
    return box_w, box_h, (outer_w // 2) - (box_w // 2), (outer_h // 2) - (box_h // 2), ratio


def iaa_letterbox_params(image_shape, new_h, new_w):
    """reference transforms.py:196-205 (``IaaLetterbox._compute_height_width_pad``, the evaluation pipeline's letterbox):
    ``(resize_w, resize_h, x_pad, y_pad)`` -- pads floored from ``(new - resized) / 2``."""
    img_h, img_w = image_shape[0:2]
    ratio = min(new_w / img_w, new_h / img_h)
    resize_w, resize_h = int(img_w * ratio), int(img_h * ratio)
    return resize_w, resize_h, (new_w - resize_w) // 2, (new_h - resize_h) // 2


def letterbox_batch(images, dim, device=None, variant="utils"):
    """List of uint8 RGB images ([H,W,3] numpy arrays or tensors, any sizes) -> network input batch
    ``[B,3,dim_h,dim_w]`` fp32 in [0,1] on the GPU + per-image transforms ``[B,5]`` (box_w, box_h, x, y, ratio).

    One HIP kernel per image does what reference utils.load_image(mode='letterbox') does on the host with cv2
    (utils.py:44-72): bicubic resize, paste on a 128-grey canvas, /255, HWC -> CHW.  ``dim`` = (w, h).

    ``variant``: ``"utils"`` (default) places the box as ``utils.letterbox_transforms`` does (``out // 2 - box // 2``);
    ``"eval"`` as the evaluation pipeline's ``IaaLetterbox`` does (``(out - box) // 2``, transforms.py:144-209 --
    ``evaluate.generate_results_file(..., is_letterbox=True)``); ``"scale"`` is ``iaa.Scale(dim)`` (evaluate.py:213): a plain
    bicubic resize to ``dim`` (transforms row: the whole canvas, ratio = None)."""
    if variant not in ("utils", "eval", "scale"):
        raise ValueError("variant must be 'utils', 'eval' or 'scale'")
    if not torch.cuda.is_available():
        raise _ffi.Yv3Error("no GPU available: this package has no CPU path")
    dev = torch.device(device if device is not None else "cuda")
    out_w, out_h = int(dim[0]), int(dim[1])
    lib = _ffi.lib()
    batch = torch.empty((len(images), 3, out_h, out_w), dtype=torch.float32, device=dev)
    trans = []
    with torch.cuda.device(dev):
        keep = []
        for b, img in enumerate(images):
            t = img if isinstance(img, torch.Tensor) else torch.from_numpy(img)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise _ffi.Yv3Error("images must be uint8 [H,W,3] RGB")
            t = t.to(dev).contiguous()
            keep.append(t)
            H, W = t.shape[0], t.shape[1]
            dst = batch.data_ptr() + b * 3 * out_h * out_w * 4
            if variant == "utils":
                _ffi.check(lib.yv3_letterbox(t.data_ptr(), H, W, dst, out_h, out_w, _ffi.stream_ptr()), "yv3_letterbox")
                trans.append(list(letterbox_transforms((W, H), (out_w, out_h))))
            else:
                bw, bh, bx, by = iaa_letterbox_params((H, W), out_h, out_w) if variant == "eval" else (out_w, out_h, 0, 0)
                _ffi.check(lib.yv3_letterbox_ex(t.data_ptr(), H, W, dst, out_h, out_w, bw, bh, bx, by, _ffi.stream_ptr()), "yv3_letterbox_ex")
                trans.append([bw, bh, bx, by, min(out_w / W, out_h / H) if variant == "eval" else float("nan")])
        torch.cuda.current_stream().synchronize()      # `keep` (device copies of the inputs) may now be released
    return batch, torch.tensor(trans, dtype=torch.float32)


def resize_batch(images, dim, device=None):
    """List of uint8 RGB images (any sizes) -> ``[B,3,dim_h,dim_w]`` fp32 in [0,1] on the GPU by a plain
    ``cv2.resize(img, dim)`` (INTER_LINEAR) + /255 + HWC -> CHW: reference utils.load_image(mode='resize')
    (utils.py:68-71), one HIP kernel per image.  ``dim`` = (w, h)."""
    if not torch.cuda.is_available():
        raise _ffi.Yv3Error("no GPU available: this package has no CPU path")
    dev = torch.device(device if device is not None else "cuda")
    out_w, out_h = int(dim[0]), int(dim[1])
    lib = _ffi.lib()
    batch = torch.empty((len(images), 3, out_h, out_w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        keep = []
        for b, img in enumerate(images):
            t = img if isinstance(img, torch.Tensor) else torch.from_numpy(img)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise _ffi.Yv3Error("images must be uint8 [H,W,3] RGB")
            t = t.to(dev).contiguous()
            keep.append(t)
            _ffi.check(lib.yv3_resize_linear(t.data_ptr(), t.shape[0], t.shape[1], batch.data_ptr() + b * 3 * out_h * out_w * 4,
                                             out_h, out_w, _ffi.stream_ptr()), "yv3_resize_linear")
        torch.cuda.current_stream().synchronize()
    return batch


def letterbox_image(img, dim):
    """reference utils.py:44-56 signature: HWC uint8 RGB ``img`` -> (HWC image with values 0..255, transform tensor)."""
    batch, trans = letterbox_batch([img], dim)
    hwc = (batch[0] * 255.0).round().permute(1, 2, 0).to(torch.int64).cpu().numpy()
    return hwc, trans[0]


def load_image(img, mode=None, dim=None):
    """reference utils.py:59-72 with the decode step left to the caller: ``img`` is a path (decoded with PIL) or
    an HWC uint8 RGB array.  Returns (CHW fp32 tensor in [0,1] on the GPU, transform or None)."""
    if isinstance(img, str):
        from PIL import Image
        import numpy as np
        img = np.asarray(Image.open(img).convert("RGB"))
    if mode == 'letterbox' and dim is not None:
        batch, trans = letterbox_batch([img], dim)
        return batch[0], trans[0]
    if mode == 'resize' and dim is not None:
        return resize_batch([img], dim)[0], None
    if mode is not None and dim is not None:
        raise ValueError("mode must be 'letterbox' or 'resize'")
    t = img if isinstance(img, torch.Tensor) else torch.from_numpy(img)
    return t.cuda().float().permute(2, 0, 1) / 255, None
