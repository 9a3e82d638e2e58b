"""Worker of tests/test_gpu_headline.py::test_config4_full_size_8_ranks_equals_single_gpu (launched under torch.distributed.run
with 8 ranks and YV3_DIST_BACKEND=gloo on a box with ONE GPU: the ranks time-share it, gloo carries the gather).

BASELINE configs[3]: 416x416, global batch 256 (seed 3), contiguous shards of 32 per rank, one gather of the final boxes.
Every rank: ``detect_sharded(net, x_global)`` with ``net.deterministic = True`` (direct one-tile-per-workgroup kernels, one lane:
the result of an image does not depend on the batch it is in), then a second call in the DEFAULT mode (per-launch kernel
choice).  Rank 0 alone then runs the single-GPU ``detect`` on all 256 images and prints one JSON line:
  bitwise_equal_images   images whose sharded result equals the single-GPU result bit for bit (deterministic mode) -- must be 256
  default_mode_unmatched_frac  set-wise delta between the default-mode sharded result and the deterministic one
  collectives_per_call   all_gather_into_tensor calls issued by one detect_sharded call -- must be 1
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch                        # noqa: E402
import torch.distributed as dist    # noqa: E402

from yolo_v3_amd import synth, detect, detect_sharded, YoloNet, WeightManager, dist as ydist   # noqa: E402
from oracle.boxdelta import boxes_delta                                                        # noqa: E402


def make_net(stream, deterministic, shared_gpu):
    net = YoloNet((416, 416)).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    net = net.cuda()
    if shared_gpu:
        net.lanes = 1              # several processes share one GPU: no second lane against each other
    if deterministic:
        net.deterministic = True
    return net


def main():
    """Two ways to run: (a) 8 ranks over gloo on ONE GPU (test_config4_full_size_8_ranks_equals_single_gpu); (b) one rank per GPU
    over RCCL on a box with >= 2 GPUs (tests/test_gpu_dist.py::test_rccl_multi_gpu_sharded_equals_single_gpu, self-arming)."""
    rank, local, world = ydist.init_from_env()
    rccl = dist.get_backend() == "nccl"
    assert (rccl and world <= torch.cuda.device_count()) or (not rccl and world == 8)
    torch.cuda.set_device(local if rccl else 0)
    stream = synth.weight_stream()
    x = torch.from_numpy(synth.images(64, 416, 3)).repeat(4, 1, 1, 1)          # 256 images on the host (64 distinct scenes x 4)
    x = x + torch.arange(256, dtype=torch.float32).view(256, 1, 1, 1) * (1.0 / 4096.0)   # ... made distinct: no two images equal
    x = x.clamp_(0.0, 1.0).contiguous()
    lo, hi = ydist.shard_range(256, rank, world)

    calls = []
    orig = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(tuple(a[1].shape)), orig(*a, **k))[1]
    net_d = make_net(stream, True, not rccl)
    with torch.no_grad():
        got_d = detect_sharded(net_d, x)                       # warm (builds the sharded detector: no collective beyond the gather at lanes=1)
        calls.clear()
        got_d = detect_sharded(net_d, x)
    per_call = len(calls)
    shapes = list(calls)
    net = make_net(stream, False, not rccl)
    with torch.no_grad():
        got = detect_sharded(net, x)
    lanes_here = [sd.det.lanes for sd in net.__dict__.get("_sharded_detectors", {}).values()]
    dist.all_gather_into_tensor = orig
    assert len(got_d) == len(got) == 256
    lanes_all = [None] * world
    dist.all_gather_object(lanes_all, lanes_here)              # (after the measured calls: which lane count every rank chose)
    if rccl:
        dist.barrier(device_ids=[local])
    else:
        dist.barrier()
    if rank == 0:
        with torch.no_grad():
            want = detect(net_d, x.cuda())                     # one GPU, all 256 images, same deterministic kernels
        assert len(want) == 256
        equal = sum(1 for a, b in zip(got_d, want) if tuple(a.shape) == tuple(b.shape) and torch.equal(a, b))
        d = boxes_delta(got, want, 256)
        print(json.dumps({"world": world, "backend": dist.get_backend(), "lanes": lanes_all,
                          "images": 256, "shard": hi - lo, "collectives_per_call": per_call, "payload": shapes,
                          "bitwise_equal_images": equal, "boxes": int(sum(b.shape[0] for b in want if b.numel())),
                          "default_mode_unmatched_frac": round(d["unmatched_frac"], 6),
                          "default_mode_max_rel_err_coords": float("%.3g" % d["max_rel_err_coords"])}))
    if rccl:
        dist.barrier(device_ids=[local])
    else:
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
