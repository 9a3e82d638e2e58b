"""``detect()``: the reference's caller idiom as one fused GPU pipeline.

The reference has no ``detect`` function; its callers write (test.py:35-36, evaluate.py:201-204)

    det1, det2, det3 = net(imgs.cuda(), None)
    detections = postprocessing(torch.cat((det1, det2, det3), 1), nc, conf, nms, is_eval, use_nms)

``detect(net, imgs, ...)`` returns exactly what that returns, but runs it as: 75 HIP conv
launches -> 3 decode launches writing the concatenated tensor directly -> filter -> rank ->
IOU masks -> scan -> compact, with no host synchronisation until ONE device-to-host copy of the
final ``[B, cap, 7]`` boxes and their counts.  ``Detector`` keeps the buffers (and optionally a
captured HIP graph of the whole pipeline) alive across calls.

Lanes.  A batch of 40 or more 416 x 416 images (19 at 608 x 608) runs as TWO contiguous sub-batches on two HIP streams (``lanes``): the same
kernels with the same K order -- bit-identical detections on the direct kernels (``net.winograd = False``); with the default
per-launch choice between the direct and the Winograd form of a 3x3 layer (it depends on the sub-batch's tile count) the two
schedules agree within fp32 round-off -- but the two launch sequences run concurrently
and fill each other's partially occupied rounds of the chip (at bs=64 the 13x13 layers have 1.34 rounds of tiles,
the 26x26 layers 2.64, ...) and overlap HBM-bound layers with matrix-bound ones: conv section 13.97 -> 12.65 ms
(tools/lanes_probe.py).  HIP maps streams onto a few hardware queues and two streams on the SAME queue serialise
(measured: 15.1 ms, worse than one lane), so the constructor looks for a stream pair that really overlaps with a deterministic
yes/no probe (`streams_run_concurrently`) -- or runs a single lane if there is none.  The lane count itself is a function of the
batch shape only (TWO_LANES_MIN_PIXELS), identical on every rank of a sharded run: no stopwatch, no collective.
"""
from collections import OrderedDict

import torch

from . import _ffi
from .engine import Plan
from .utils import PostProcessor, boxes_to_list


class Detector:
    def __init__(self, net, batch, height, width, obj_conf_thr=0.5, nms_thr=0.4, is_eval=False, use_nms=True,
                 max_cand=None, cap=None, dtype=None, graph=False, lanes=None, group=None, sync_lanes=False):
        """lanes: 1, 2, or None = automatic: 2 from `two_lanes_min_pixels(mode)` (default mode: 40 images of 416 x 416; exact fp32 52; BF16 120) upwards when a stream pair that
        really runs concurrently exists (a deterministic probe, see `concurrent_stream_pair`), else 1 -- a pure function of the
        batch shape, no stopwatch; ``net.lanes`` overrides the default.
        group / sync_lanes: accepted for `ShardedDetector`; the constructor contains NO collective (round 5: the lane count is
        the same function of the shape on every rank)."""
        self.net = net
        self.shape = (batch, 3, height, width)
        self.conf, self.nms_thr, self.is_eval, self.use_nms = obj_conf_thr, nms_thr, is_eval, use_nms
        self.engine = net.engine(dtype)
        self.engine.ensure_packed()
        self.device = self.engine.device
        self._generation = self.engine.generation
        self._group, self._sync_lanes = group, bool(sync_lanes)
        if lanes is None:
            from .engine import measure_env
            lanes = getattr(net, "lanes", None) or (int(measure_env("YV3_LANES")) if measure_env("YV3_LANES") else None)
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
            auto2 = lanes is None and batch >= 2 and batch * height * width >= two_lanes_min_pixels(self.engine.dtype)
            if auto2:
                self._choose_lanes()
            else:
                self._build_plans(2 if (lanes or 1) >= 2 else 1, self.engine.lane_choices.get("stream_pair"))
                if self.lanes == 2:
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
        """Automatic lane count: a function of the batch shape -- two lanes from TWO_LANES_MIN_PIXELS upwards -- provided a
        stream pair that really runs concurrently exists on this GPU (`concurrent_stream_pair`: a yes/no probe, not a stopwatch, but
        still an observation of this GPU at this moment: a rank whose GPU is busy with something else may answer "no", run one lane
        and say so in a warning and in `bench.py`'s per-rank table).  No collective: the shape rule is the same on every rank, the
        final choice is per rank."""
        streams = concurrent_stream_pair(self.device, self.engine.lane_choices)
        if streams is None and not self.engine.lane_choices.get("one_lane_warned"):
            import warnings
            self.engine.lane_choices["one_lane_warned"] = True
            warnings.warn("yolo_v3_amd: no pair of HIP streams ran concurrently on %s when this Detector was built (GPU shared with another "
                          "process?): a batch of this size normally runs as two lanes (+4...9 %%), this one runs as one; the probe is repeated "
                          "by the next Detector of this network (up to three times)" % (self.device,), RuntimeWarning)
        self._build_plans(2 if streams is not None else 1, streams)

    def _pick_stream_pair(self):
        """Two lanes were REQUESTED: they must not land on one hardware queue (they would serialise: 6.0 instead of 4.2 ms at bs=16,
        profiles/r03y_lanes_loop_probe.txt).  Keeps the freshly made streams when no concurrent pair exists."""
        pair = concurrent_stream_pair(self.device, self.engine.lane_choices)
        if pair is not None:
            self.lane_streams = pair

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

    def recover(self, err):
        """A kernel status word came back set.  StreamKTimeout: the engine drops the stream-K schedule; RangeOverflow in the default
        mode: this detector moves to the fall-back engine (F32X3: same planes pipeline, fp32 range; `engine.range_fallback`).  True when
        the batch can be run again (the next `run_device` rebuilds the plans), False when the error stands."""
        from .engine import StreamKTimeout, RangeOverflow
        if isinstance(err, StreamKTimeout):
            self.engine.disable_stream_k()
            return True
        if isinstance(err, RangeOverflow):
            fb = self.engine.range_fallback()
            if fb is None:
                return False
            fb.ensure_packed()
            self.engine = fb
            self._generation = None                  # run_device rebuilds plans (and the graph) from the new engine
            return True
        return False

    def run_checked(self, imgs):
        """run_device + fetch + status check -> (host counts, boxes); a stream-K hand-over time-out (the GPU was shared with another
        small-batch caller, engine.StreamKTimeout) switches the schedule off for this engine, a value beyond the fp16 range in the
        default mode (engine.RangeOverflow) moves the detector to F32X3, and the batch is run again."""
        from .engine import StreamKTimeout, RangeOverflow
        for attempt in (0, 1, 2):
            boxes, counts = self.run_device(imgs)
            host_counts, bx, status = self.fetch(boxes, counts)
            try:
                self.engine.raise_if_overflowed(self.plan, status)
                return host_counts, bx
            except (StreamKTimeout, RangeOverflow) as e:
                if attempt == 2 or not self.recover(e):
                    raise

    def __call__(self, imgs):
        host_counts, bx = self.run_checked(imgs)
        return self.to_list(bx, host_counts)


# Two lanes from this many input pixels per batch (40 images of 416 x 416, 19 of 608 x 608).  Round 5, with the four-wave conv tile in the
# one-lane plans (tools/lanes_threshold_check.sh, profiles/r05ai_two_lanes_threshold_check.txt; bench.py, lanes forced, alternating): 416x416 bs=32
# -3.5 %, bs=36 -3 %, bs=40 +4 %, bs=48 +4.7 %, bs=56 +9 %; 608x608 bs=16 -4 %, bs=20 (43 equivalents) +2 %, bs=24 +6.6 %.  Earlier, same box, whole pipeline,
# sustained (profiles/r03y_lane_choice.txt, r03y_lane_choice_order.txt): bs=64 @416 +7 %, 256 images +6 %; bs=32 @416, bs=16 @416,
# bs=16 @608 (bf16) and the dense bs=8 @608 config: -1...-4 % to +2 % depending on the run -- a wash that a 1 s stopwatch calibration
# decided differently from box to box.  The rule is the documented threshold; `lanes=` / ``net.lanes`` override it.
TWO_LANES_MIN_PIXELS = 40 * 416 * 416
# ... per math mode (images of 416 x 416; same measurement, profiles/r05ai_two_lanes_threshold_check.txt): the default fp16-plane mode and the
# bf16x3 mode cross over at 40 (f32x3 bs=40 +5 %); exact fp32 at ~52 (bs=48 -0.9 %, bs=56 +2.4 %, bs=64 +6.2 %); BF16 only beyond ~112
# (bs=64 ONE lane +6.8 %, bs=96 / 112 equal, bs=128 two lanes +3.5 %; 608x608 bs=32 one lane +6 %)
TWO_LANES_MIN_IMAGES_416 = {_ffi.F32H2: 40, _ffi.F32X3: 40, _ffi.F32: 52, _ffi.BF16: 120}


def two_lanes_min_pixels(dtype):
    """Input pixels per batch from which `Detector` runs two lanes in math mode `dtype` (a pure function of the mode and the batch shape)."""
    return TWO_LANES_MIN_IMAGES_416.get(dtype, 40) * 416 * 416


def streams_run_concurrently(a, b, device):
    """Deterministic yes/no probe: does work on stream `b` start while stream `a` is busy?  HIP maps streams onto a few hardware
    queues; two streams on the SAME queue serialise.  A spin of a few milliseconds goes to `a`, one tiny kernel to `b`: when the tiny
    kernel has finished and the spin has not, the two streams are on different queues."""
    with torch.cuda.device(device):
        t = torch.zeros(64, device=device)
        torch.cuda.synchronize(device)
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(a):
            torch.cuda._sleep(8_000_000)            # >= 3 ms at any of the clocks the counter may run at (100 MHz ... 2.4 GHz)
            ea.record(a)
        with torch.cuda.stream(b):
            t.add_(1.0)
            eb.record(b)
        eb.synchronize()
        concurrent = not ea.query()
        ea.synchronize()
    return concurrent


def concurrent_stream_pair(device, cache=None, tries=6):
    """Two HIP streams of `device` that run concurrently (None when none of `tries` fresh streams pairs with the first);
    remembered in `cache["stream_pair"]` (the engine's lane_choices: one probe per engine)."""
    if cache is not None and cache.get("stream_pair") is not None:
        return cache["stream_pair"]
    # a NEGATIVE answer is not remembered for ever: the probe watches wall time, and a GPU that was busy with another
    # process when the first detector was built can make a tiny kernel miss the spin.  Up to three constructions re-probe; after
    # that the engine keeps its single lane (with one warning from `Detector._choose_lanes`)
    if cache is not None:
        if cache.get("stream_pair_misses", 0) >= 3:
            return None
    pair = None
    if not hasattr(torch.cuda, "_sleep"):            # (a torch build without the spin kernel: two fresh streams, unprobed)
        pair = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
        if cache is not None:
            cache["stream_pair"] = pair
        return pair
    with torch.cuda.device(device):
        first = torch.cuda.Stream(device=device)
        for _ in range(tries):
            other = torch.cuda.Stream(device=device)
            if streams_run_concurrently(first, other, device) and streams_run_concurrently(other, first, device):
                pair = [first, other]
                break
    if cache is not None:
        if pair is None:
            cache["stream_pair_misses"] = cache.get("stream_pair_misses", 0) + 1
        else:
            cache["stream_pair"] = pair
    return pair


def _detector_cache(net):
    """At most DETECTOR_CACHE_MAX detectors (buffers of ~10 GB each at bs=64) per net, least recently used first out:
    alternating between two batch shapes (a last partial batch) does not re-allocate or re-calibrate."""
    return net.__dict__.setdefault("_detectors", OrderedDict())


DETECTOR_CACHE_MAX = 3


def cached_detector(net, key, build, sharded=False):
    """`sharded`: detectors of `detect_sharded` live in their OWN least-recently-used cache: it only ever sees the sharded calls,
    which all ranks issue in the same order with the same keys, so the rank-local `detect()` / `predict()` calls of one rank cannot
    push a sharded detector (its payload buffers are the collective's registered memory) out on that rank alone."""
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
        host_counts, bx = det.run_checked(imgs)
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
