"""RCCL on the GPU that is available: a world_size-1 ``nccl`` (= RCCL on ROCm) process group with the final-box
all-gather FORCED (no world==1 early return), so librccl is loaded and the collective of yolo_v3_amd/dist.py really
executes on device.  The N>1 plumbing (uneven shards, padding, ordering) is covered on CPU by tests/test_dist_gloo.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from yolo_v3_amd import synth, detect, detect_sharded, dist as ydist
from tests.helpers import load_sw1_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def test_gather_boxes_runs_on_rccl(nccl_world1):
    boxes = torch.arange(3 * 5 * 7, dtype=torch.float32, device="cuda").view(3, 5, 7)
    counts = torch.tensor([2, 0, 5], dtype=torch.int32, device="cuda")
    same_b, same_c = ydist.gather_boxes(boxes, counts)                   # world of one: identity, no collective
    assert same_b is boxes and same_c is counts
    gb, gc = ydist.gather_boxes(boxes, counts, force=True)               # the collective itself, on device
    torch.cuda.synchronize()
    assert gb.data_ptr() != boxes.data_ptr() and torch.equal(gb, boxes) and torch.equal(gc, counts)
    assert "librccl" in open("/proc/self/maps").read(), "RCCL was not loaded by the nccl backend"


def test_detect_sharded_equals_detect(nccl_world1, sw1_stream):
    """The sharded product path (shard -> Detector.run_device -> RCCL all-gather -> list conversion) returns exactly
    what the single-GPU ``detect`` returns, bit for bit; BASELINE configs[3]'s per-GPU share (32 x 416 x 416)."""
    net = load_sw1_net(sw1_stream).cuda()
    x = torch.from_numpy(synth.images(32, 416, 3))                      # config 4: seed 3, 32 images per rank
    want = detect(net, x.cuda())
    got = detect_sharded(net, x, force_collective=True)                  # global batch on the host, shard moved to the GPU
    assert len(got) == len(want) == 32
    for a, b in zip(got, want):
        assert tuple(a.shape) == tuple(b.shape) and torch.equal(a, b)
    got_local = detect_sharded(net, x.cuda(), local_shard=True, force_collective=True)
    for a, b in zip(got_local, want):
        assert torch.equal(a, b)
    assert sum(len(b) for b in want) > 32


def test_sharded_path_issues_one_collective(nccl_world1, sw1_stream):
    """The product's sharded step = Detector.run_device + pack + exactly ONE all_gather_into_tensor carrying boxes, counts
    and the status word; the gathered payload unpacks to the detector's own boxes / counts bit for bit."""
    net = load_sw1_net(sw1_stream).cuda()
    x = torch.from_numpy(synth.images(4, 416, 9)).cuda()
    sd = ydist.ShardedDetector(net, 4, 416, 416, cap=256, force_collective=True)
    calls = []
    orig = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1]
    try:
        gathered = sd.run_device(x)
        torch.cuda.synchronize()
    finally:
        dist.all_gather_into_tensor = orig
    assert calls == [torch.Size([4, 257, 7])]
    boxes, meta = ydist.unpack_payload(gathered)
    assert torch.equal(boxes, sd.det.boxes[:, :256]) and torch.equal(meta[:, 0], sd.det.counts[:4]) and torch.equal(meta[:, 1], sd.det.counts[4:])
    assert int(meta[:, 2].max()) == 0
    res = sd.assemble(gathered, [(0, 4)])
    want = detect(net, x)
    assert all(torch.equal(a, b) for a, b in zip(res, want))


def test_bench_two_ranks_on_this_gpu_over_gloo():
    """bench.py's N > 1 path end to end on the one GPU of this box: `python bench.py --gpus 2` re-launches itself as two ranks
    (YV3_DIST_BACKEND=gloo: the ranks share the GPU, gloo carries the gather), each runs the product's sharded step
    (`ShardedDetector.run_device`: pipeline -> pack -> ONE all-gather) and rank 0 prints one JSON line with n_gpus = 2 -- the
    launch line, rendezvous, shard seeds, collective, max-over-ranks timing and `assemble` of the driver's multi-GPU runs."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["YV3_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "3", "--warmup", "1",
                        "--lanes", "1", "--no-extras"], cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["entry"].startswith("ShardedDetector.run_device") and "[8, 513, 7]" in out["config"]["collective"]
    assert len(r.stdout.strip().splitlines()[-1]) < 4096                        # the compact line the driver parses
    full = json.load(open(os.path.join(repo, out["full"])))                     # everything else: the side file
    assert full["n_gpus"] == 2 and full["value"] == out["value"]
    assert out["value"] > 0 and "gather" in full["stages_ms"] and len(full["config"]["boxes_kept_first_images"]) == 4
    assert [r_["lanes"] for r_ in out["ranks"]] == [1, 1]


def test_rccl_multi_gpu_sharded_equals_single_gpu():
    """SELF-ARMING: on a box with >= 2 GPUs -- the driver's scaling node; the build's own boxes have one -- spawn
    one rank per GPU (up to 8) over RCCL and run BASELINE configs[3] (416x416, global batch 256, contiguous shards) through
    ``detect_sharded``: every image bit-identical to the single-GPU ``detect`` under ``net.deterministic``, the default-mode result within
    the set-wise tolerance, exactly ONE collective per call, and the SAME lane count on every rank.  Skipped on one GPU, where
    tests/test_gpu_headline.py::test_config4_full_size_8_ranks_equals_single_gpu runs the same worker as 8 gloo ranks."""
    import json
    import subprocess
    import sys
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n < 2:
        pytest.skip("needs >= 2 GPUs (found %d): the N > 1 RCCL path is exercised by the driver's multi-GPU node" % n)
    world = max(w for w in (2, 4, 8) if w <= n)                  # 256 images divide evenly
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "YV3_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "tests", "dist_config4_worker.py")]
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print("config 4 over RCCL, %d GPUs:" % world, out)
    assert out["world"] == world and out["backend"] == "nccl" and out["images"] == 256 and out["shard"] == 256 // world
    assert out["collectives_per_call"] == 1 and out["bitwise_equal_images"] == 256 and out["boxes"] > 2000
    assert out["default_mode_unmatched_frac"] <= 0.002
    assert len(out["lanes"]) == world and all(l == out["lanes"][0] for l in out["lanes"]), out["lanes"]


def test_yv3_gather_boxes_is_one_rccl_allgather():
    """The C entry point of the path's one exchange (include/yv3.h: yv3_gather_boxes) on a communicator the CALLER owns:
    a world-1 ncclComm_t made with RCCL's own API (ctypes on the librccl torch ships), the payload `pack_payload` builds, one
    ncclAllGather on the current stream -- the gathered tensor must equal the payload bit for bit and unpack to the inputs."""
    import ctypes
    from yolo_v3_amd import _ffi
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0 and comm.value
    try:
        B, cap = 5, 9
        boxes = torch.rand(B, cap, 7, device="cuda")
        cand = torch.arange(B, dtype=torch.int32, device="cuda") * 3
        kept = torch.arange(B, dtype=torch.int32, device="cuda")
        payload = ydist.pack_payload(boxes, cand, kept, torch.tensor([4], dtype=torch.int32, device="cuda"))
        out = torch.zeros_like(payload)
        lib = _ffi.lib()
        _ffi.check(lib.yv3_gather_boxes(payload.data_ptr(), out.data_ptr(), B, cap + 1, comm, _ffi.stream_ptr()), "yv3_gather_boxes")
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), payload.view(torch.int32))
        gb, meta = ydist.unpack_payload(out)
        assert torch.equal(gb, boxes) and meta[:, 0].tolist() == cand.tolist() and meta[:, 1].tolist() == kept.tolist() and set(meta[:, 2].tolist()) == {4}
        assert lib.yv3_gather_boxes(payload.data_ptr(), out.data_ptr(), B, cap + 1, None, _ffi.stream_ptr()) == -1     # YV3_EINVAL: no communicator
    finally:
        rccl.ncclCommDestroy(comm)
