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
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_boxes(boxes, counts, group=None):
    """All-gather equal-shaped shards.

    boxes  [B_local, cap, 7] float32, counts [B_local] int32 (number of valid rows per image).
    Returns (boxes [world*B_local, cap, 7], counts [world*B_local]) in rank order on every rank.
    """
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
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
