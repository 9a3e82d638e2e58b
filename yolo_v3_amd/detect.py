"""``detect()``: the reference's caller idiom as one fused GPU pipeline.

The reference has no ``detect`` function; its callers write (test.py:35-36, evaluate.py:201-204)

    det1, det2, det3 = net(imgs.cuda(), None)
    detections = postprocessing(torch.cat((det1, det2, det3), 1), nc, conf, nms, is_eval, use_nms)

``detect(net, imgs, ...)`` returns exactly what that returns, but runs it as: 75 HIP conv
launches -> 3 decode launches writing the concatenated tensor directly -> filter -> rank ->
IOU masks -> scan -> compact, with no host synchronisation until ONE device-to-host copy of the
final ``[B, cap, 7]`` boxes and their counts.  ``Detector`` keeps the buffers (and optionally a
captured HIP graph of the whole pipeline) alive across calls.
"""
import torch

from . import _ffi
from .utils import PostProcessor


class Detector:
    def __init__(self, net, batch, height, width, obj_conf_thr=0.5, nms_thr=0.4, is_eval=False, use_nms=True,
                 max_cand=None, cap=None, dtype=None, graph=False):
        self.net = net
        self.shape = (batch, 3, height, width)
        self.conf, self.nms_thr, self.is_eval, self.use_nms = obj_conf_thr, nms_thr, is_eval, use_nms
        self.engine = net.engine(dtype)
        self.engine.ensure_packed()
        self.device = self.engine.device
        self._generation = self.engine.generation
        with torch.cuda.device(self.device):
            self.plan = self.engine.plan(batch, height, width)
            self.dets = torch.empty((batch, self.plan.N, self.plan.attrib), device=self.device, dtype=torch.float32)
            n = self.plan.N
            self.pp = PostProcessor(batch, n, net.numClass, self.device,
                                    max_cand=max_cand or (min(n * net.numClass, 16384) if is_eval else n), cap=cap)
        self._graph = None
        self._static_in = None
        self._want_graph = graph
        self.boxes = None

    # -- pipeline pieces (all asynchronous on the current stream)
    def _enqueue(self, x, mark=None):
        """`mark(name)`, if given, is called at every stage boundary (bench.py records a HIP event on the launch
        stream there: per-stage split conv0 / convs / decode / filter / nms)."""
        mark = mark or (lambda name: None)
        mark("start")
        self.engine.run_front(self.plan, x)
        mark("conv0")
        self.engine.run_conv_sequence(self.plan, self.dets)
        mark("convs")
        self.engine.run_decode(self.plan, self.dets)
        mark("decode")
        # scores are sigmoid products: PP_PROB lets the filter skip rows whose objectness already fails
        self.pp.filter(self.dets, self.conf, self.is_eval, prob=True)
        mark("filter")
        self.boxes = self.pp.nms(self.dets, self.nms_thr, self.use_nms, self.pp.max_cand, self.pp.cap)
        mark("nms")

    def _capture(self, x):
        self._static_in = torch.empty_like(x)
        self._static_in.copy_(x)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                       # warm-up outside capture (lazy module loads, workspaces)
                self._enqueue(self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue(self._static_in)
        self._graph = g

    def run_device(self, imgs, mark=None):
        """Enqueue one batch; returns (boxes [B,cap,7], counts [2B]) still on the GPU, no sync."""
        x = self.engine.prepare_input(imgs)
        if tuple(x.shape) != self.shape:
            raise _ffi.Yv3Error("Detector was built for %s, got %s" % (self.shape, tuple(x.shape)))
        with torch.cuda.device(self.device):
            self.engine.ensure_packed()
            if self.engine.generation != self._generation:      # parameters changed: packed weights / plan were rebuilt
                self._generation = self.engine.generation
                self.plan = self.engine.plan(self.shape[0], self.shape[2], self.shape[3])
                self._graph = None
            if self._want_graph:
                if self._graph is None:
                    self._capture(x)
                self._static_in.copy_(x)
                self._graph.replay()
            else:
                self._enqueue(x, mark)
        return self.boxes, self.pp.counts

    def __call__(self, imgs):
        boxes, counts = self.run_device(imgs)
        host = torch.cat((counts, self.plan.flags)).cpu()    # the single D2H sync: counts + saturation flag
        self.engine.raise_if_overflowed(self.plan, int(host[-1]))
        return self.pp.to_list(boxes, host[:-1])


def detect(net, imgs, num_classes=None, obj_conf_thr=0.5, nms_thr=0.4, is_eval=False, use_nms=True):
    """``postprocessing(torch.cat(net(imgs, None), 1), ...)`` fused on the GPU (see module docstring)."""
    if num_classes is not None and num_classes != net.numClass:
        raise _ffi.Yv3Error("num_classes=%d does not match net.numClass=%d" % (num_classes, net.numClass))
    _ffi.require_cuda(imgs, "imgs")
    if is_eval:
        # multi-label mode can produce up to N*C candidates per image: size the buffers from the
        # actual counts (one extra host sync) instead of the worst case
        from .utils import postprocessing
        with torch.no_grad():
            return postprocessing(net.forward_cat(imgs), net.numClass, obj_conf_thr, nms_thr, True, use_nms)
    key = (tuple(imgs.shape), imgs.device, float(obj_conf_thr), float(nms_thr), bool(is_eval), bool(use_nms), net.math_mode)
    cache = net.__dict__.setdefault("_detectors", {})
    det = cache.get(key)
    if det is None:
        cache.clear()                                        # keep at most one set of buffers alive
        det = cache[key] = Detector(net, imgs.shape[0], imgs.shape[2], imgs.shape[3], obj_conf_thr, nms_thr, is_eval, use_nms)
    with torch.no_grad():
        return det(imgs)


def predict(net, images, dim=(416, 416), num_classes=None, obj_conf_thr=0.5, nms_thr=0.4, is_letterbox=True):
    """The whole of reference test.py:28-46 (``predict``) for a list of uint8 RGB images of any size:
    GPU letterbox (or, ``is_letterbox=False``, plain resize) -> network -> post-processing -> boxes mapped back to
    each ORIGINAL image.

    Returns one ``[n_i, 5]`` CPU tensor per image: ``cls, x, y, w, h`` (original-image pixels, clipped), exactly
    the rows ``torch.cat((prediction[..., 6:7], correct_yolo_boxes(...)), -1)`` of the reference; images without
    detections give an empty tensor."""
    from .utils import letterbox_batch, resize_batch
    from . import boundingbox
    # is_letterbox=False: the images were brought to `dim` by a plain resize (load_image mode='resize', utils.py:68-69)
    # and the boxes go back through rescale_bbox instead of letterbox_reverse (test.py:41, boundingbox.py:139-149)
    batch = letterbox_batch(images, dim)[0] if is_letterbox else resize_batch(images, dim)
    res = detect(net, batch, num_classes, obj_conf_thr, nms_thr)
    out = []
    for i, img in enumerate(images):
        pred = res[i] if i < len(res) else torch.Tensor()
        if pred.numel() == 0:
            out.append(torch.zeros((0, 5)))
            continue
        org_h, org_w = int(img.shape[0]), int(img.shape[1])
        xywh = boundingbox.correct_yolo_boxes(pred[:, 0:4], org_w, org_h, int(dim[0]), int(dim[1]), bool(is_letterbox))
        out.append(torch.cat((pred[:, 6:7], xywh), -1))
    return out
