"""N>1 path on CPU: world_size-2, -4 and -8 gloo runs (8: three ranks own NO image of the 5-image batch) of the shard + final-box all-gather (yolo_v3_amd/dist.py)."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolo_v3_amd import dist as ydist


def test_shard_range_partitions():
    for n, w in [(256, 8), (10, 4), (3, 8), (64, 1)]:
        spans = [ydist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = ydist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    B, cap = 3, 5
    lo, hi = ydist.shard_range(B * world, rank, world)
    boxes = torch.zeros(B, cap, 7)
    counts = torch.zeros(B, dtype=torch.int32)
    for i, img in enumerate(range(lo, hi)):
        counts[i] = img % (cap + 1)
        boxes[i, :counts[i]] = float(img) + torch.arange(int(counts[i]))[:, None] / 10.0
    ab, ac = ydist.gather_boxes(boxes, counts)
    ok = ab.shape == (B * world, cap, 7) and ac.tolist() == [i % (cap + 1) for i in range(B * world)]
    for img in range(B * world):
        n = int(ac[img])
        ok = ok and bool((ab[img, :n, 0] == float(img) + torch.arange(n) / 10.0).all())
    lst = ydist.boxes_to_list(ab, ac)
    ok = ok and len(lst) == B * world and lst[0].shape == (0,) and lst[1].shape == (1, 7)
    # the sharded-detect plumbing around the single-GPU pipeline (dist.detect_sharded): UNEVEN split of a global
    # batch of 5 over the ranks, padded shards, one all-gather, padding dropped, reference result convention
    Bg, cap2 = 5, 4
    imgs = torch.arange(Bg, dtype=torch.float32).view(Bg, 1, 1, 1).expand(Bg, 3, 2, 2).contiguous()     # image id in every pixel
    x, b_pad, spans = ydist.take_shard(imgs, rank, world)
    want_spans = [ydist.shard_range(Bg, r_, world) for r_ in range(world)]      # world 2: [(0, 3), (3, 5)]; world 4: 2, 1, 1, 1 images; world 8: five ranks with one image, three with none
    want_pad = max(h_ - l_ for l_, h_ in want_spans)
    ok = ok and b_pad == want_pad and [tuple(sp) for sp in spans] == want_spans and x.shape[0] == want_pad
    ids = x[:, 0, 0, 0].to(torch.int64)                                 # what this rank "detects": image g keeps g % 3 boxes
    bx = torch.zeros(b_pad, cap2, 7)
    meta = torch.zeros(b_pad, 3, dtype=torch.int32)
    for i, g in enumerate(ids.tolist()):
        n = g % 3
        meta[i, 0], meta[i, 1] = n, n
        bx[i, :n, 0] = float(g)
        bx[i, :n, 6] = torch.arange(n, dtype=torch.float32)
    # the product composition: pack (boxes + one int32 meta row per image) -> ONE all-gather -> assemble
    calls = []
    orig = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    payload = ydist.pack_payload(bx, meta[:, 0], meta[:, 1], torch.zeros(1, dtype=torch.int32))
    gathered = ydist.gather_payload(payload)
    dist.all_gather_into_tensor = orig
    ok = ok and len(calls) == 1 and gathered.shape == (world * b_pad, cap2 + 1, 7)
    res, status = ydist.assemble_global(gathered, spans, b_pad, max_cand=10)
    ok = ok and status == 0 and len(res) == Bg
    for g in range(Bg):
        n = g % 3
        ok = ok and (tuple(res[g].shape) == ((n, 7) if n else (0,)))
        ok = ok and (n == 0 or bool((res[g][:, 0] == float(g)).all()))
    # nothing anywhere -> the [] sentinel; a status bit set on ONE rank is seen by all
    zero = torch.zeros(b_pad, dtype=torch.int32)
    payload0 = ydist.pack_payload(torch.zeros(b_pad, cap2, 7), zero, zero, torch.tensor([rank], dtype=torch.int32))   # rank r reports status r
    res0, status0 = ydist.assemble_global(ydist.gather_payload(payload0), spans, b_pad, 10)
    want_status = 0
    for r_ in range(world):
        want_status |= r_
    ok = ok and res0 == [] and status0 == want_status
    # rank-consistent lane count WITHOUT a collective: the automatic choice is a pure function of the batch shape
    import importlib
    ydet = importlib.import_module("yolo_v3_amd.detect")           # (the package exports the FUNCTION detect under that name)
    ok = ok and ydet.TWO_LANES_MIN_PIXELS == 40 * 416 * 416 and not hasattr(ydet, "_min_over_group")
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gather_boxes_gloo(world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


def test_single_process_is_identity():
    b, c = torch.zeros(2, 4, 7), torch.tensor([1, 0], dtype=torch.int32)
    ab, ac = ydist.gather_boxes(b, c)
    assert ab is b and ac is c
    assert ydist.boxes_to_list(torch.zeros(2, 4, 7), torch.zeros(2, dtype=torch.int32)) == []
