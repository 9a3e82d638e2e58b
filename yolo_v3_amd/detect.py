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
kernels with the same K order -- bit-identical detections on the direct kernels (``net.winograd = False``); with the default
per-launch choice between the direct and the Winograd form of a 3x3 layer (it depends on the sub-batch's tile count) the two
schedules agree within fp32 round-off -- but the two launch sequences run concurrently
and fill each other's partially occupied rounds of the chip (at bs=64 the 13x13 layers have 1.34 rounds of tiles,
the 26x26 layers 2.64, ...) and overlap HBM-bound layers with matrix-bound ones: conv section 13.97 -> 12.65 ms
(tools/lanes_probe.py).  HIP maps streams onto a few hardware queues and two streams on the SAME queue serialise
(measured: 15.1 ms, worse than one lane), so the constructor times the candidate stream pairs on this GPU and
keeps the fastest -- or a single lane if none wins.
"""
import time
from collections import OrderedDict

import torch
import torch.distributed as dist

from . import _ffi
from .engine import Plan
from .utils import PostProcessor, boxes_to_list


class Detector:
    def __init__(self, net, batch, height, width, obj_conf_thr=0.5, nms_thr=0.4, is_eval=False, use_nms=True,
                 max_cand=None, cap=None, dtype=None, graph=False, lanes=None, group=None, sync_lanes=False):
        """lanes: 1, 2, or None = automatic (2 when the batch is at least ~12 images of 416 x 416 and a stream pair that really
        runs concurrently is found, see the module docstring; ``net.lanes`` / YV3_LANES override the default).
        sync_lanes: all ranks of the torch.distributed process group `group` (None = the default group) construct this
        Detector together (`detect_sharded`); the automatic lane decision is then MIN-reduced over them so that every
        rank runs the same lane count."""
        self.net = net
        self.shape = (batch, 3, height, width)
        self.conf, self.nms_thr, self.is_eval, self.use_nms = obj_conf_thr, nms_thr, is_eval, use_nms
        self.engine = net.engine(dtype)
        self.engine.ensure_packed()
        self.device = self.engine.device
        self._generation = self.engine.generation
        self._group, self._sync_lanes = group, bool(sync_lanes)
        if lanes is None:
            import os
            lanes = getattr(net, "lanes", None) or (int(os.environ["YV3_LANES"]) if os.environ.get("YV3_LANES") else None)
        if self.engine.deterministic:
            lanes = 1                    # net.deterministic: one schedule whatever the batch (engine.Engine.__init__)
        self._lanes_req = lanes
        with torch.cuda.device(self.device):
            B = batch
            n, attrib = Plan.geometry(self.engine, height, width)        # (rows N, attributes) without allocating a plan
            self.N = n
            self.max_cand = int(max_cand or (min(n * net.numClass, 16384) if is_eval else n))
            self.cap = int(cap or self.max_cand)
            self.dets = torch.empty((B, n, attrib), device=self.device, dtype=torch.float32)
            # ONE result tensor + ONE counts buffer ([0:B] candidates, [B:2B] kept) for all lanes: a single D2H copy
            self.boxes = torch.empty((B, self.cap, 7), device=self.device, dtype=torch.float32)
            self.counts = torch.zeros(2 * B, device=self.device, dtype=torch.int32)
            # automatic: worth trying from ~12 images of 416 x 416 upwards (measured crossover; 608 x 608: 8), calibration decides
            auto2 = lanes is None and batch >= 2 and batch * height * width >= 12 * 416 * 416
            self.lane_calibration = None
            if auto2:
                self._choose_lanes()
            else:
                self._build_plans(2 if (lanes or 1) >= 2 else 1, self.engine.lane_choices.get("stream_pair"))
                if self.lanes == 2 and "stream_pair" not in self.engine.lane_choices:
                    self._pick_stream_pair()              # (a requested second lane must not land on the first one's hardware queue)
        self._graph = None
        self._static_in = None
        self._want_graph = graph

    # -- lanes
    def _build_plans(self, lanes, streams=None):
        """Per lane: a Plan (activation buffers + descriptors) for its contiguous sub-batch, a HIP stream and a
        PostProcessor whose outputs are views of the Detector's result tensor / counts buffer."""
        B, _, H, W = self.shape
        lanes = max(1, min(int(lanes), B))
        self.lanes = lanes
        nc = self.net.numClass
        if lanes == 1:
            self.plan = self.engine.plan(B, H, W)                 # the engine's cached plan (shared with net.forward)
            self.lane_plans, self.lane_off, self.lane_streams = [self.plan], [0], []
        else:
            sizes = [B // lanes + (1 if i < B % lanes else 0) for i in range(lanes)]
            flags = torch.zeros(1, device=self.device, dtype=torch.int32)
            self.lane_plans = [Plan(self.engine, b, H, W, flags=flags, two_lanes=lanes == 2) for b in sizes]     # own buffers per lane, ONE status word
            self.lane_off = [sum(sizes[:i]) for i in range(lanes)]
            self.lane_streams = streams or [torch.cuda.Stream(device=self.device) for _ in range(lanes)]
            self.plan = self.lane_plans[0]
        self.lane_pp = [PostProcessor(p.B, self.N, nc, self.device, max_cand=self.max_cand, cap=self.cap,
                                      counts=(self.counts[off:off + p.B], self.counts[B + off:B + off + p.B]),
                                      out=self.boxes[off:off + p.B])
                        for p, off in zip(self.lane_plans, self.lane_off)]

    def _lane_body(self, p, pp, xi, di, mark, post=True):
        """One lane's whole pipeline on the current stream: front -> convs (+ decode) -> filter -> NMS."""
        self.engine.run_front(p, xi)
        mark("conv0")
        self.engine.run_conv_sequence(p, di)
        mark("convs")
        self.engine.run_decode(p, di)
        mark("decode")
        if not post:
            return
        # scores are sigmoid products: PP_PROB lets the filter skip rows whose objectness already fails
        pp.filter(di, self.conf, self.is_eval, prob=True)
        mark("filter")
        pp.nms(di, self.nms_thr, self.use_nms, self.max_cand, self.cap)
        mark("nms")

    def _run_lanes(self, x, mark, post=True):
        """Every lane's pipeline; returns on the current stream with all lanes joined.  With two lanes each lane runs its
        OWN filter + NMS on its own stream into its image range of the shared result tensor (no join before the
        post-processing).  Measured neutral against post-processing the joined batch (the 8-wave conv workgroups
        fill every CU's register file, so NMS kernels only run in the gaps between conv launches whatever the stream layout);
        the stage marks of a multi-lane step are 'conv0' = fork, 'convs' ... 'nms' = join (all of it is the concurrent
        section)."""
        noop = lambda name: None
        if self.lanes == 1:
            self._lane_body(self.plan, self.lane_pp[0], x, self.dets, mark, post)
            return
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        mark("conv0")
        for p, pp, off, st in zip(self.lane_plans, self.lane_pp, self.lane_off, self.lane_streams):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                self._lane_body(p, pp, x[off:off + p.B], self.dets[off:off + p.B], noop, post)
                done = torch.cuda.Event()
                done.record(st)
            main.wait_event(done)
        for name in ("convs", "decode", "filter", "nms"):
            mark(name)

    def _choose_lanes(self):
        """Automatic lane count: time the candidate stream pairs against one lane on this GPU (streams that share a
        hardware queue serialise and LOSE against a single lane) -- once per (engine, batch shape): the decision and the
        winning stream pair are cached on the engine -- then agree on MIN over the ranks of `group`."""
        key = ("lanes", self.shape, self.is_eval)
        cached = self.engine.lane_choices.get(key)
        if cached is None:
            cached = self._calibrate_lanes()
            self.engine.lane_choices[key] = cached
        lanes, streams, self.lane_calibration = cached
        if self._sync_lanes:
            lanes = _min_over_group(lanes, self._group, self.device)
        self._build_plans(lanes, streams if lanes > 1 else None)

    def _pick_stream_pair(self):
        """Two-lane plans are built: time three candidate stream pairs over a few steps each and keep the fastest (a pair that shares
        a hardware queue serialises: 6.0 instead of 4.2 ms at bs=16, profiles/r03y_lanes_loop_probe.txt); remembered on the engine."""
        x = torch.rand(self.shape, device=self.device, dtype=torch.float32, generator=torch.Generator(device=self.device).manual_seed(1234))
        noop = lambda name: None
        best = None
        for pair in [self.lane_streams] + [[torch.cuda.Stream(device=self.device) for _ in range(2)] for _ in range(2)]:
            self.lane_streams = pair
            for _ in range(2):
                self._run_lanes(x, noop)
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(4):
                self._run_lanes(x, noop)
            torch.cuda.synchronize(self.device)
            t = time.perf_counter() - t0
            if best is None or t < best[0]:
                best = (t, pair)
        self.lane_streams = best[1]
        self.lane_plans[0].flags.zero_()
        self.engine.lane_choices["stream_pair"] = best[1]

    def _calibrate_lanes(self):
        B, _, H, W = self.shape
        # image-like values, not zeros: the conv kernels' speed is data-dependent on this power-limited chip (zero operands run
        # 25-30 % faster and favour the wrong schedule)
        gen = torch.Generator(device=self.device).manual_seed(1234)
        x = torch.rand(self.shape, device=self.device, dtype=torch.float32, generator=gen)
        noop = lambda name: None

        reps = 3 if B * H * W >= 40 * 416 * 416 else 8           # short steps need more repetitions for a 3 % decision

        def timed(n=reps):
            # the WHOLE pipeline (convs + decode + filter + NMS): two lanes also run two smaller, latency-bound post-processing
            # sequences -- timing the convolutions alone chose two lanes for bs=16 and the dense 608x608 bs=8 config, where the
            # full step is 3-4 % slower with them (profiles/r03y_lane_choice.txt)
            for _ in range(2):
                self._run_lanes(x, noop)
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(n):
                self._run_lanes(x, noop)
            torch.cuda.synchronize(self.device)
            return (time.perf_counter() - t0) / n

        def snapshot():
            return (self.lanes, self.plan, self.lane_plans, self.lane_off, self.lane_streams, self.lane_pp)

        def restore(st):
            self.lanes, self.plan, self.lane_plans, self.lane_off, self.lane_streams, self.lane_pp = st

        def steady(seconds):
            """ms per step over at least `seconds` of back-to-back steps (host sync every 4 steps)."""
            self._run_lanes(x, noop)
            torch.cuda.synchronize(self.device)
            n, t0 = 0, time.perf_counter()
            while True:
                for _ in range(4):
                    self._run_lanes(x, noop)
                n += 4
                torch.cuda.synchronize(self.device)
                if time.perf_counter() - t0 >= seconds:
                    return (time.perf_counter() - t0) / n

        # 1. the stream pair: short runs are enough to tell a pair that shares a hardware queue (it serialises) from one that does not
        known = self.engine.lane_choices.get("stream_pair")
        self._build_plans(2, known)
        if known is None:
            best = None
            for pair in [self.lane_streams] + [[torch.cuda.Stream(device=self.device) for _ in range(2)] for _ in range(2)]:
                self.lane_streams = pair
                t = timed()
                if best is None or t < best[0]:
                    best = (t, pair)
            known = self.engine.lane_choices["stream_pair"] = best[1]
        self.lane_streams = known
        best = (None, known)
        two_state = snapshot()
        had_plan = (B, H, W) in self.engine._plans
        self._build_plans(1)
        one_state = snapshot()
        # 2. one lane or two: alternating runs of >= 0.15 s each.  A burst of a few steps runs at boost clocks; the sustained
        # clock under the power limit is lower for the schedule that keeps more of the chip busy, and bursts chose two lanes for
        # bs=16 / the dense 608x608 bs=8 config where the sustained step is 3-8 % slower with them (profiles/r03y_lane_choice.txt)
        t1 = t2 = 0.0
        for _ in range(2):
            restore(one_state)
            t1 += steady(0.15) / 2
            restore(two_state)
            t2 += steady(0.15) / 2
        one_state[1].flags.zero_()
        two_state[2][0].flags.zero_()
        # two lanes must win by 4 %: inside a caller's loop (D2H copies of the results between the steps) they lose 1-3 % of what
        # this loop measures and their step time scatters more (profiles/r03y_lane_choice_order.txt: bs=16 and the dense config are a
        # wash at a measured 2-3.5 % advantage; bs=64 wins 7.5 % here and 6-7 % in the bench loop)
        two = t2 < 0.96 * t1
        restore(one_state)
        if two and not had_plan:
            self.engine.drop_plan(B, H, W)            # the one-lane plan's activation buffers are not kept alive beside the lanes'
        del one_state, two_state
        return (2 if two else 1, best[1] if two else None,
                {"one_lane_ms": round(t1 * 1e3, 3), "two_lanes_ms": round(t2 * 1e3, 3)})

    # -- pipeline pieces (all asynchronous on the current stream)
    def _enqueue(self, x, mark=None):
        """`mark(name)`, if given, is called at every stage boundary (bench.py records a HIP event on the launch
        stream there: per-stage split conv0 / convs / decode / filter / nms)."""
        mark = mark or (lambda name: None)
        mark("start")
        self._run_lanes(x, mark)

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
                self._build_plans(self.lanes, self.lane_streams or None)     # keep the calibrated pair
                self._graph = None
            if self._want_graph:
                if self._graph is None:
                    self._capture(x)
                self._static_in.copy_(x)
                self._graph.replay()
            else:
                self._enqueue(x, mark)
        return self.boxes, self.counts

    def to_list(self, boxes, counts_host):
        return boxes_to_list(boxes, counts_host, self.shape[0], self.max_cand)

    HOST_CAP, HOST_CAP_EVAL = 512, 4096      # kept boxes per image that travel with the first (and normally only) D2H copy

    def fetch(self, boxes, counts):
        """The path's single host synchronisation: counts + status word + the first HOST_CAP box rows of every image go to
        pinned host buffers with asynchronous copies behind the kernels, then ONE stream sync.  Returns (host counts [2B],
        host boxes or -- when an image kept more than HOST_CAP boxes -- the device tensor, status word)."""
        B = self.shape[0]
        hc = min(self.cap, self.HOST_CAP_EVAL if self.is_eval else self.HOST_CAP)
        if getattr(self, "_host_meta", None) is None:
            self._host_meta = torch.empty(2 * B + 1, dtype=torch.int32).pin_memory()
            self._host_boxes = torch.empty((B, hc, 7), dtype=torch.float32).pin_memory()
        with torch.cuda.device(self.device):
            self._host_meta[:2 * B].copy_(counts, non_blocking=True)
            self._host_meta[2 * B:].copy_(self.plan.flags, non_blocking=True)
            self._host_boxes.copy_(boxes[:, :hc], non_blocking=True)
            torch.cuda.current_stream().synchronize()
        meta = self._host_meta.clone()
        fits = int(meta[B:2 * B].max()) <= hc
        return meta[:2 * B], (self._host_boxes if fits else boxes), int(meta[2 * B])

    def __call__(self, imgs):
        boxes, counts = self.run_device(imgs)
        host_counts, bx, status = self.fetch(boxes, counts)
        self.engine.raise_if_overflowed(self.plan, status)
        return self.to_list(bx, host_counts)


def _min_over_group(value, group, device):
    """MIN of an int over the ranks of `group` (on the backend's device: RCCL wants GPU tensors, gloo CPU ones)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.tensor([int(value)], dtype=torch.int32, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def _detector_cache(net):
    """At most DETECTOR_CACHE_MAX detectors (buffers of ~10 GB each at bs=64) per net, least recently used first out:
    alternating between two batch shapes (a last partial batch) does not re-allocate or re-calibrate."""
    return net.__dict__.setdefault("_detectors", OrderedDict())


DETECTOR_CACHE_MAX = 3


def cached_detector(net, key, build, sharded=False):
    """`sharded`: detectors of `detect_sharded` live in their OWN least-recently-used cache.  Building one contains a
    collective (the MIN-reduction of the lane count), so every rank must build -- and therefore evict -- at the same calls:
    this cache only ever sees the sharded calls, which all ranks issue in the same order with the same keys, whereas the
    rank-local `detect()` / `predict()` calls of one rank can no longer push a sharded detector out on that rank alone."""
    cache = net.__dict__.setdefault("_sharded_detectors", OrderedDict()) if sharded else _detector_cache(net)
    det = cache.get(key)
    if det is None:
        while len(cache) >= DETECTOR_CACHE_MAX:
            cache.popitem(last=False)
        det = cache[key] = build()
    else:
        cache.move_to_end(key)
    return det


def detect(net, imgs, num_classes=None, obj_conf_thr=0.5, nms_thr=0.4, is_eval=False, use_nms=True):
    """``postprocessing(torch.cat(net(imgs, None), 1), ...)`` fused on the GPU (see module docstring).

    Eval mode (``is_eval=True``: every (row, class) pair above the threshold is a candidate, up to N*C per image) runs the
    same fused `Detector` with room for 16 384 candidates per image; a batch that exceeds it is re-run through the
    two-phase path (`forward_cat` -> `postprocessing`, buffers sized from the actual counts, one more host sync)."""
    if num_classes is not None and num_classes != net.numClass:
        raise _ffi.Yv3Error("num_classes=%d does not match net.numClass=%d" % (num_classes, net.numClass))
    _ffi.require_cuda(imgs, "imgs")
    key = (tuple(imgs.shape), imgs.device, float(obj_conf_thr), float(nms_thr), bool(is_eval), bool(use_nms), net.math_mode)
    det = cached_detector(net, key, lambda: Detector(net, imgs.shape[0], imgs.shape[2], imgs.shape[3], obj_conf_thr, nms_thr,
                                                     is_eval, use_nms))
    with torch.no_grad():
        if not is_eval:
            return det(imgs)
        boxes, counts = det.run_device(imgs)
        host_counts, bx, status = det.fetch(boxes, counts)
        det.engine.raise_if_overflowed(det.plan, status)
        B = imgs.shape[0]
        if int(host_counts[:B].max()) <= det.max_cand and int(host_counts[B:2 * B].max()) <= det.cap:
            return det.to_list(bx, host_counts)
        from .utils import postprocessing
        return postprocessing(det.dets, net.numClass, obj_conf_thr, nms_thr, True, use_nms)


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
