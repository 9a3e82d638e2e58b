"""Data-parallel sharding of the image batch over the GPUs of one node.

The reference is single-GPU (no torch.distributed / DataParallel anywhere).  Images are
independent on this path (convs, decode and NMS are all per image; utils.py:152 loops
``for batch_idx in range(nB)``), so scaling out is: one process per GPU, contiguous batch shards,
replicated weights, and ONE exchange at the very end -- an all-gather of the fixed-capacity
``[B_local, cap, 7]`` box tensor and the per-image counts over RCCL/xGMI (``backend="nccl"`` is
RCCL on ROCm).  ~0.1-1 MB per rank: latency-bound, no ring all-reduce anywhere.
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
    """All-gather equal-shaped shards.

    boxes  [B_local, cap, 7] float32, counts [B_local] int32 (number of valid rows per image).
    Returns (boxes [world*B_local, cap, 7], counts [world*B_local]) in rank order on every rank.
    A world of one returns its inputs; ``force=True`` runs the collective even then (exercises RCCL on a
    single GPU: tests/test_gpu_dist.py).
    """
    if not dist.is_available() or not dist.is_initialized():
        return boxes, counts
    if dist.get_world_size(group) == 1 and not force:
        return boxes, counts
    world = dist.get_world_size(group)
    all_boxes = torch.empty((world * boxes.shape[0],) + tuple(boxes.shape[1:]), dtype=boxes.dtype, device=boxes.device)
    all_counts = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(all_boxes, boxes.contiguous(), group=group)      # concatenated along dim 0
    dist.all_gather_into_tensor(all_counts, counts.contiguous(), group=group)
    return all_boxes, all_counts


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


def assemble_global(all_boxes, all_meta, spans, b_pad, max_cand, cap):
    """Gathered ``[world*b_pad, cap, 7]`` boxes + ``[world*b_pad, 3]`` int32 meta rows (candidates, kept, status) ->
    the reference's result convention for the GLOBAL batch (list of ``[n,7]`` CPU tensors in image order, an empty
    ``torch.Tensor()`` for an image without candidates, ``[]`` when no image has any: utils.py:153-158,248-251).
    Returns (result, status word OR-ed over all ranks)."""
    from . import _ffi
    all_meta = all_meta.view(-1, 3).cpu()
    status = 0
    for v in all_meta[:, 2].tolist():
        status |= int(v)
    keep = []
    for r, (l, h) in enumerate(spans):
        keep += list(range(r * b_pad, r * b_pad + (h - l)))
    if not keep:
        return [], status
    ncand, nkeep = all_meta[keep, 0].tolist(), all_meta[keep, 1].tolist()
    if max(ncand) > max_cand:
        raise _ffi.Yv3Error("candidate buffer overflow (%d > %d)" % (max(ncand), max_cand))
    if max(nkeep) > cap:
        raise _ffi.Yv3Error("more than cap=%d boxes kept for one image (%d): raise cap" % (cap, max(nkeep)))
    if sum(ncand) == 0:
        return [], status
    host = all_boxes[:, :max(max(nkeep), 1)].cpu()
    return [host[i, :nkeep[j]].clone() if ncand[j] else torch.Tensor() for j, i in enumerate(keep)], status


def detect_sharded(net, imgs, num_classes=None, obj_conf_thr=0.5, nms_thr=0.4, use_nms=True, group=None,
                   local_shard=False, cap=512, dtype=None, force_collective=False):
    """``detect(net, imgs, ...)`` over all ranks of the process group: the product form of BASELINE configs[3]
    (416x416 bs=256 over 8 MI355X).

    Every rank passes the SAME global batch ``imgs`` [B,3,H,W] (CPU or GPU; only the rank's contiguous shard
    ``shard_range(B, rank, world)`` is moved to its GPU and run) -- or, with ``local_shard=True``, its own shard
    of equal size on every rank.  Each rank runs the fused single-GPU pipeline (`Detector.run_device`: 75 convs ->
    decode -> filter -> NMS, no host sync) on its shard; the ONLY exchange is one all-gather of the fixed-capacity
    ``[B_local, cap, 7]`` boxes + candidate / kept counts over RCCL/xGMI (`gather_boxes`).  Returns, on every
    rank, exactly what ``detect`` returns for the global batch: the reference's list of per-image ``[n,7]`` CPU
    tensors in global image order, or ``[]`` (test.py:35-36 / utils.py:248).  Without an initialised process group
    it is ``detect`` on one GPU.  ``cap`` bounds the kept boxes per image that travel (overflow raises).
    """
    from . import _ffi
    from .detect import Detector
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
    key = ("sharded", b_pad, tuple(x.shape[1:]), float(obj_conf_thr), float(nms_thr), bool(use_nms), int(cap), dtype, net.math_mode)
    cache = net.__dict__.setdefault("_detectors", {})
    det = cache.get(key)
    if det is None:
        cache.clear()
        det = cache[key] = Detector(net, b_pad, x.shape[2], x.shape[3], obj_conf_thr, nms_thr, False, use_nms, cap=cap, dtype=dtype)
    with torch.no_grad(), torch.cuda.device(dev):
        boxes, counts = det.run_device(x)
        cap_ = min(cap, boxes.shape[1])
        # one int32 row per image: candidates, kept, the status word (fp16 saturation flag) of this rank's kernels
        meta = torch.stack((counts[:b_pad], counts[b_pad:], det.plan.flags.expand(b_pad)), 1).contiguous()
        all_boxes, all_meta = gather_boxes(boxes[:, :cap_].contiguous(), meta.view(-1), group, force_collective)
        result, status = assemble_global(all_boxes, all_meta, spans, b_pad, det.pp.max_cand, cap_)   # the single host sync
    det.engine.raise_if_overflowed(det.plan, status)
    return result
