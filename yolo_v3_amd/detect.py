"""``detect()``: the reference's caller idiom as one fused GPU pipeline.

The reference has no ``detect`` function; its callers write (test.py:35-36, evaluate.py:201-204)

    det1, det2, det3 = net(imgs.cuda(), None)
    detections = postprocessing(torch.cat((det1, det2, det3), 1), nc, conf, nms, is_eval, use_nms)

``detect(net, imgs, ...)`` returns exactly what that returns, but runs it as: 75 HIP conv
launches -> 3 decode launches writing the concatenated tensor directly -> filter -> rank ->
IOU masks -> scan -> compact, with no host synchronisation until ONE device-to-host copy of the
final ``[B, cap, 7]`` boxes and their counts.  ``Detector`` keeps the buffers (and optionally a
captured HIP graph of the whole pipeline) alive across calls.

Lanes.  A batch of about 12 or more 416 x 416 images (8 at 608 x 608) runs as TWO contiguous sub-batches on two HIP streams (``lanes``): the same
kernels with the same K order, so the detections are bit-identical, but the two launch sequences run concurrently
and fill each other's partially occupied rounds of the chip (at bs=64 the 13x13 layers have 1.34 rounds of tiles,
the 26x26 layers 2.64, ...) and overlap HBM-bound layers with matrix-bound ones: conv section 13.97 -> 12.65 ms
(tools/lanes_probe.py).  HIP maps streams onto a few hardware queues and two streams on the SAME queue serialise
(measured: 15.1 ms, worse than one lane), so the constructor times the candidate stream pairs on this GPU and
keeps the fastest -- or a single lane if none wins.
"""
import time

import torch

from . import _ffi
from .engine import Plan
from .utils import PostProcessor


class Detector:
    def __init__(self, net, batch, height, width, obj_conf_thr=0.5, nms_thr=0.4, is_eval=False, use_nms=True,
                 max_cand=None, cap=None, dtype=None, graph=False, lanes=None):
        """lanes: 1, 2, or None = automatic (2 when the batch is at least ~12 images of 416 x 416 and a stream pair that really
        runs concurrently is found, see the module docstring; ``net.lanes`` / YV3_LANES override the default)."""
        self.net = net
        self.shape = (batch, 3, height, width)
        self.conf, self.nms_thr, self.is_eval, self.use_nms = obj_conf_thr, nms_thr, is_eval, use_nms
        self.engine = net.engine(dtype)
        self.engine.ensure_packed()
        self.device = self.engine.device
        self._generation = self.engine.generation
        if lanes is None:
            import os
            lanes = getattr(net, "lanes", None) or (int(os.environ["YV3_LANES"]) if os.environ.get("YV3_LANES") else None)
        self._lanes_req = lanes
        with torch.cuda.device(self.device):
            # automatic: worth trying from ~12 images of 416 x 416 upwards (measured crossover; 608 x 608: 8), calibration decides
            auto2 = lanes is None and batch >= 2 and batch * height * width >= 12 * 416 * 416
            self._build_plans(2 if auto2 or (lanes or 1) >= 2 else 1)
            self.dets = torch.empty((batch, self.plan.N, self.plan.attrib), device=self.device, dtype=torch.float32)
            n = self.plan.N
            self.pp = PostProcessor(batch, n, net.numClass, self.device,
                                    max_cand=max_cand or (min(n * net.numClass, 16384) if is_eval else n), cap=cap)
            if self.lanes > 1 and lanes is None:
                self._calibrate_lanes()
        self._graph = None
        self._static_in = None
        self._want_graph = graph
        self.boxes = None

    # -- lanes
    def _build_plans(self, lanes):
        B, _, H, W = self.shape
        lanes = max(1, min(int(lanes), B))
        self.lanes = lanes
        if lanes == 1:
            self.plan = self.engine.plan(B, H, W)                 # the engine's cached plan (shared with net.forward)
            self.lane_plans, self.lane_off, self.lane_streams = [self.plan], [0], []
            return
        sizes = [B // lanes + (1 if i < B % lanes else 0) for i in range(lanes)]
        flags = torch.zeros(1, device=self.device, dtype=torch.int32)
        self.lane_plans = [Plan(self.engine, b, H, W, flags=flags) for b in sizes]     # own buffers per lane, ONE status word
        self.lane_off = [sum(sizes[:i]) for i in range(lanes)]
        self.lane_streams = [torch.cuda.Stream(device=self.device) for _ in range(lanes)]
        self.plan = self.lane_plans[0]

    def _run_convs(self, x, mark):
        """conv0 ... head convs (+ decode) of every lane; returns on the current stream with all lanes joined."""
        if self.lanes == 1:
            self.engine.run_front(self.plan, x)
            mark("conv0")
            self.engine.run_conv_sequence(self.plan, self.dets)
            mark("convs")
            self.engine.run_decode(self.plan, self.dets)
            return
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        mark("conv0")                                             # (lanes: the front kernels are part of the 'convs' stage)
        for p, off, st in zip(self.lane_plans, self.lane_off, self.lane_streams):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                xi, di = x[off:off + p.B], self.dets[off:off + p.B]
                self.engine.run_front(p, xi)
                self.engine.run_conv_sequence(p, di)
                self.engine.run_decode(p, di)
                done = torch.cuda.Event()
                done.record(st)
            main.wait_event(done)
        mark("convs")

    def _calibrate_lanes(self):
        """Keep the stream pair that really runs the two lanes concurrently on this GPU (streams that share a hardware queue
        serialise and LOSE against a single lane); fall back to one lane if no candidate beats it."""
        B, _, H, W = self.shape
        x = torch.zeros(self.shape, device=self.device, dtype=torch.float32)
        noop = lambda name: None

        def timed(n=3):
            for _ in range(2):
                self._run_convs(x, noop)
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(n):
                self._run_convs(x, noop)
            torch.cuda.synchronize(self.device)
            return (time.perf_counter() - t0) / n

        two_plans, two_off = self.lane_plans, self.lane_off
        cands = [self.lane_streams] + [[torch.cuda.Stream(device=self.device) for _ in range(2)] for _ in range(2)]
        best = None
        for pair in cands:
            self.lane_streams = pair
            t = timed()
            if best is None or t < best[0]:
                best = (t, pair)
        self._build_plans(1)
        t1 = timed()
        self.lane_calibration = {"one_lane_ms": round(t1 * 1e3, 3), "two_lanes_ms": round(best[0] * 1e3, 3)}
        if best[0] < 0.97 * t1:
            self.lanes, self.lane_plans, self.lane_off, self.lane_streams, self.plan = 2, two_plans, two_off, best[1], two_plans[0]
        self.plan.flags.zero_()

    # -- pipeline pieces (all asynchronous on the current stream)
    def _enqueue(self, x, mark=None):
        """`mark(name)`, if given, is called at every stage boundary (bench.py records a HIP event on the launch
        stream there: per-stage split conv0 / convs / decode / filter / nms)."""
        mark = mark or (lambda name: None)
        mark("start")
        self._run_convs(x, mark)
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
            if self.engine.generation != self._generation:      # parameters changed: packed weights / plans were rebuilt
                self._generation = self.engine.generation
                streams = self.lane_streams
                self._build_plans(self.lanes)
                if streams:
                    self.lane_streams = streams                 # keep the calibrated pair
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
