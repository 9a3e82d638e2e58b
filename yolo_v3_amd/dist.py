"""Data-parallel sharding of the image batch over the GPUs of one node.

The reference is single-GPU (no torch.distributed / DataParallel anywhere).  Images are
independent on this path (convs, decode and NMS are all per image; utils.py:152 loops
``for batch_idx in range(nB)``), so scaling out is: one process per GPU, contiguous batch shards,
replicated weights, and ONE exchange at the very end -- a single ``all_gather_into_tensor`` of the
fixed-capacity ``[B_local, cap + 1, 7]`` payload (``cap`` box rows + one row carrying the image's
int32 candidate count, kept count and the rank's kernel status word, bit-cast into the fp32 row)
over RCCL/xGMI (``backend="nccl"`` is RCCL on ROCm).  ~0.1-1 MB per rank: latency-bound, no ring
all-reduce anywhere.
The same code runs on CPU tensors with the ``gloo`` backend (tests/test_dist_gloo.py).
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of ``n_items`` for ``rank``; the first ``n % world`` ranks get one extra."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            # YV3_DIST_BACKEND=gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (RCCL wants one GPU per rank)
            backend = os.environ.get("YV3_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_boxes(boxes, counts, group=None, force=False):
    """All-gather equal-shaped shards of boxes ``[B_local, cap, 7]`` + int32 ``counts [B_local]`` (or ``[B_local*k]``
    flattened rows of k <= 7 int32 per image) with ONE collective: the counts ride in an extra box row (`pack_payload`'s
    layout).  Returns (boxes [world*B_local, cap, 7], counts [world*B_local*k]) in rank order on every rank.  A world
    of one returns its inputs; ``force=True`` runs the collective even then (tests/test_gpu_dist.py)."""
    if not dist.is_available() or not dist.is_initialized():
        return boxes, counts
    if dist.get_world_size(group) == 1 and not force:
        return boxes, counts
    B, cap = boxes.shape[0], boxes.shape[1]
    k = counts.numel() // max(B, 1)
    assert k * B == counts.numel() and 1 <= k <= 7, "counts must hold 1..7 int32 per image"
    payload = torch.zeros((B, cap + 1, 7), dtype=torch.float32, device=boxes.device)
    payload[:, :cap].copy_(boxes)
    payload[:, cap].view(torch.int32)[:, :k].copy_(counts.view(B, k))
    out = gather_payload(payload, group, force)
    return out[:, :cap].contiguous(), out[:, cap].view(torch.int32)[:, :k].reshape(-1).contiguous()


def boxes_to_list(boxes, counts, cand_counts=None):
    """[B,cap,7] + counts -> the reference's list-of-tensors convention (utils.py:148-202,248-251)."""
    boxes, counts = boxes.cpu(), counts.cpu().tolist()
    cand = cand_counts.cpu().tolist() if cand_counts is not None else counts
    if sum(cand) == 0:
        return []
    return [boxes[b, :counts[b]].clone() if cand[b] else torch.Tensor() for b in range(len(counts))]


def shard_plan(n_images, world, local_shard=False):
    """(padded shard size, [(lo, hi) of the global images each rank owns]).  ``local_shard``: ``n_images`` is the
    per-rank count and every rank owns exactly that many."""
    if local_shard:
        return n_images, [(r * n_images, (r + 1) * n_images) for r in range(world)]
    return (n_images + world - 1) // world, [shard_range(n_images, r, world) for r in range(world)]


def take_shard(imgs, rank, world, local_shard=False):
    """This rank's images, padded to the common shard size with copies of its last image (equal shapes for the
    all-gather; the padding rows are dropped again by `assemble_global`)."""
    b_pad, spans = shard_plan(imgs.shape[0], world, local_shard)
    x = imgs if local_shard else imgs[spans[rank][0]:spans[rank][1]]
    if x.shape[0] < b_pad:
        fill = x[-1:] if x.shape[0] else torch.zeros((1,) + tuple(imgs.shape[1:]), dtype=imgs.dtype, device=imgs.device)
        x = torch.cat((x, fill.expand(b_pad - x.shape[0], *([-1] * (imgs.dim() - 1)))), 0)
    return x, b_pad, spans


META_FIELDS = 3          # int32 per image in the payload's last row: candidates, kept, status word


def pack_payload(boxes, cand_counts, kept_counts, status, payload=None):
    """``boxes [B, >=cap, 7]`` fp32 + per-image int32 counts + the rank's status word (0-dim / 1-element int32 tensor)
    -> ``payload [B, cap+1, 7]`` fp32: rows ``[0, cap)`` are the boxes, row ``cap`` holds the three int32 values
    bit-cast into its first three floats.  ``payload`` (pre-allocated) fixes ``cap``; all on the tensors' device."""
    B = boxes.shape[0]
    if payload is None:
        payload = torch.empty((B, boxes.shape[1] + 1, 7), dtype=torch.float32, device=boxes.device)
    cap = payload.shape[1] - 1
    payload[:, :cap].copy_(boxes[:, :cap])
    meta = payload[:, cap].view(torch.int32)              # [B, 7] int32 view of the last row (same storage)
    meta[:, 0].copy_(cand_counts)
    meta[:, 1].copy_(kept_counts)
    meta[:, 2].copy_(status.reshape(-1)[:1].expand(B))
    meta[:, 3:].zero_()
    return payload


def unpack_payload(payload):
    """-> (boxes view ``[B, cap, 7]`` fp32, meta ``[B, 3]`` int32: candidates, kept, status)."""
    cap = payload.shape[1] - 1
    return payload[:, :cap], payload[:, cap].view(torch.int32)[:, :META_FIELDS]


def gather_payload(payload, group=None, force=False, out=None):
    """THE collective of the sharded path: one ``all_gather_into_tensor`` of the equal-shaped payloads, rank order along
    dim 0.  A world of one returns its input (``force=True`` runs the collective even then: tests/test_gpu_dist.py)."""
    if not dist.is_available() or not dist.is_initialized():
        return payload
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return payload
    if out is None:
        out = torch.empty((world * payload.shape[0],) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(out, payload.contiguous(), group=group)
    return out


def assemble_global(gathered, spans, b_pad, max_cand):
    """Gathered ``[world*b_pad, cap+1, 7]`` payload -> the reference's result convention for the GLOBAL batch (list of
    ``[n,7]`` CPU tensors in image order, an empty ``torch.Tensor()`` for an image without candidates, ``[]`` when no
    image has any: utils.py:153-158,248-251).  The host copy of the meta rows is the path's single host sync.
    Returns (result, status word OR-ed over all ranks)."""
    from . import _ffi
    boxes, meta = unpack_payload(gathered)
    cap = boxes.shape[1]
    meta = meta.cpu()
    status = 0
    for v in meta[:, 2].tolist():
        status |= int(v)
    keep = []
    for r, (l, h) in enumerate(spans):
        keep += list(range(r * b_pad, r * b_pad + (h - l)))
    if not keep:
        return [], status
    ncand, nkeep = meta[keep, 0].tolist(), meta[keep, 1].tolist()
    if max(ncand) > max_cand:
        raise _ffi.Yv3Error("candidate buffer overflow (%d > %d)" % (max(ncand), max_cand))
    if max(nkeep) > cap:
        raise _ffi.Yv3Error("more than cap=%d boxes kept for one image (%d): raise cap" % (cap, max(nkeep)))
    if sum(ncand) == 0:
        return [], status
    host = boxes[:, :max(max(nkeep), 1)].cpu()
    return [host[i, :nkeep[j]].clone() if ncand[j] else torch.Tensor() for j, i in enumerate(keep)], status


class ShardedDetector:
    """This rank's share of a sharded detection: the fused single-GPU `Detector` for ``b_pad`` images + the payload
    buffers of the final gather.  `run_device` enqueues everything up to and including the collective and returns the
    gathered payload on the GPU (no host sync); `assemble` turns it into the reference's list (the one host sync).
    `detect_sharded` is ``assemble(run_device(shard))``; `bench.py --gpus N` times `run_device` (+ the D2H copy)."""

    def __init__(self, net, b_pad, height, width, obj_conf_thr=0.5, nms_thr=0.4, use_nms=True, cap=512, dtype=None,
                 group=None, force_collective=False, lanes=None):
        from .detect import Detector
        self.group, self.force = group, force_collective
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        # no collective in here: the automatic lane count is a function of the batch shape (detect.TWO_LANES_MIN_PIXELS), the same
        # on every rank; a rank without a concurrent stream pair runs one lane (same boxes within fp32 round-off, `net.deterministic`
        # for identical bits) and says so in bench.py's per-rank table
        self.det = Detector(net, b_pad, height, width, obj_conf_thr, nms_thr, False, use_nms, cap=cap, dtype=dtype,
                            lanes=lanes, group=group, sync_lanes=on and self.world > 1)
        self.b_pad = b_pad
        self.cap = min(int(cap), self.det.cap)
        dev = self.det.device
        self.payload = torch.empty((b_pad, self.cap + 1, 7), dtype=torch.float32, device=dev)
        self.gathered = (torch.empty((self.world * b_pad, self.cap + 1, 7), dtype=torch.float32, device=dev)
                         if (self.world > 1 or force_collective) else None)

    def run_device(self, x, mark=None):
        det = self.det
        boxes, counts = det.run_device(x, mark)
        with torch.cuda.device(det.device):
            pack_payload(boxes, counts[:self.b_pad], counts[self.b_pad:], det.plan.flags, self.payload)
            out = gather_payload(self.payload, self.group, self.force, self.gathered)
            if mark is not None:
                mark("gather")
        return out

    def assemble(self, gathered, spans):
        result, status = assemble_global(gathered, spans, self.b_pad, self.det.max_cand)
        self.det.engine.raise_if_overflowed(self.det.plan, status)
        return result


def detect_sharded(net, imgs, num_classes=None, obj_conf_thr=0.5, nms_thr=0.4, use_nms=True, group=None,
                   local_shard=False, cap=512, dtype=None, force_collective=False):
    """``detect(net, imgs, ...)`` over all ranks of the process group: the product form of BASELINE configs[3]
    (416x416 bs=256 over 8 MI355X).

    Every rank passes the SAME global batch ``imgs`` [B,3,H,W] (CPU or GPU; only the rank's contiguous shard
    ``shard_range(B, rank, world)`` is moved to its GPU and run) -- or, with ``local_shard=True``, its own shard
    of equal size on every rank.  Each rank runs the fused single-GPU pipeline (`Detector.run_device`: 75 convs ->
    decode -> filter -> NMS, no host sync) on its shard; the ONLY exchange is ONE all-gather of the fixed-capacity
    ``[B_local, cap+1, 7]`` payload (boxes + a row with the candidate / kept counts and the status word) over
    RCCL/xGMI.  Returns, on every rank, exactly what ``detect`` returns for the global batch: the reference's list of
    per-image ``[n,7]`` CPU tensors in global image order, or ``[]`` (test.py:35-36 / utils.py:248).  Without an
    initialised process group it is ``detect`` on one GPU.  ``cap`` bounds the kept boxes per image that travel
    (overflow raises).  All ranks must call it together (it contains collectives)."""
    from . import _ffi
    from .detect import cached_detector
    if num_classes is not None and num_classes != net.numClass:
        raise _ffi.Yv3Error("num_classes=%d does not match net.numClass=%d" % (num_classes, net.numClass))
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    dev = next(net.parameters()).device
    if dev.type != "cuda":
        raise _ffi.Yv3Error("YoloNet parameters must live on this rank's GPU (net.cuda())")
    x, b_pad, spans = take_shard(imgs, rank, world, local_shard)
    x = x.to(dev, non_blocking=True).float().contiguous()
    key = ("sharded", b_pad, tuple(x.shape[1:]), float(obj_conf_thr), float(nms_thr), bool(use_nms), int(cap), dtype,
           net.math_mode, bool(force_collective), group)        # (the group OBJECT: an id() can be reused after a group is collected)
    sd = cached_detector(net, key, lambda: ShardedDetector(net, b_pad, x.shape[2], x.shape[3], obj_conf_thr, nms_thr, use_nms,
                                                           cap, dtype, group, force_collective), sharded=True)
    from .engine import StreamKTimeout, RangeOverflow
    with torch.no_grad():
        for attempt in (0, 1, 2):
            try:
                return sd.assemble(sd.run_device(x), spans)               # assemble: the single host sync
            except (StreamKTimeout, RangeOverflow) as e:
                # the status word is OR-ed over the ranks' payloads: EVERY rank sees the same error and takes the same branch, so the
                # repeated call (one more collective) stays symmetric (the sharded path used to raise where detect() recovers)
                if attempt == 2 or not sd.det.recover(e):
                    raise
