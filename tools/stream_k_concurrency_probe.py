"""Two small-batch callers sharing one GPU: two nets (own engines, plans, stream-K workspaces) drive `Detector`s of `BB` images on two
CONCURRENT streams.  Stream-K launches of both interleave on the chip, so neither has all of its workgroups resident at once.  Round 4's
schedule (a tile's head waits for parts held by HIGHER workgroup indices) could then wait on a workgroup that was not dispatched yet, until
the hand-over timed out; round 5's (dumped part first, collectors wait for LOWER indices only) cannot stall.  Checks: every result equals the
solo result bit for bit, no status bit, the schedule is still on.   BB=2 N=300 python tools/stream_k_concurrency_probe.py
MODE=f32: the same for the exact-fp32 mode, whose F(4x4,3x3) stage cuts the items of small launches into ranges of patch rows (round 6): a range
with an item's first row waits for parts of workgroups dispatched right behind it, which deposit their part before anything else."""
import importlib, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import synth, Detector, _ffi
from tests.helpers import load_sw1_net
ydet = importlib.import_module("yolo_v3_amd.detect")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
B, N = int(os.environ.get("BB", "2")), int(os.environ.get("N", "300"))
stream = synth.weight_stream()
nets = [load_sw1_net(stream).cuda() for _ in range(2)]
xs = [torch.from_numpy(synth.images(B, 416, 11 + i)).cuda() for i in range(2)]
F32 = os.environ.get("MODE") == "f32"
dets = [Detector(n, B, 416, 416, lanes=1, **({"dtype": _ffi.F32} if F32 else {})) for n in nets]
if F32:
    assert any(f == 2 for d in dets for _, f in d.plan.forms()), "no F(4x4) launch at this batch size"
    assert any(n == 2 for d in dets for n in d.plan.launches())
else:
    assert all(d.plan.workspace is not None for d in dets), "stream-K is not on at this batch size"
solo = [d(x) for d, x in zip(dets, xs)]
pair = ydet.concurrent_stream_pair(dev, {})
assert pair is not None
hosts = [torch.empty((B, 512, 7)).pin_memory() for _ in range(2)]
hc = [torch.empty((2 * B,), dtype=torch.int32).pin_memory() for _ in range(2)]
bad = 0
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    t0 = time.perf_counter()
    for it in range(N):
        for i in range(2):
            with torch.cuda.stream(pair[i]):
                b, c = dets[i].run_device(xs[i])
                hosts[i].copy_(b[:, :512], non_blocking=True); hc[i].copy_(c, non_blocking=True)
        if it % 10 == 9:
            torch.cuda.synchronize()
            for i in range(2):
                got = dets[i].to_list(hosts[i], hc[i])
                if not (len(got) == len(solo[i]) and all(torch.equal(a, b_) for a, b_ in zip(got, solo[i]))):
                    bad += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
flags = [int(d.plan.flags.item()) for d in dets]
print("two concurrent %s callers, bs=%d each, %d iterations: %.3f ms per pair of calls; results differing from solo: %d; status words %s; "
      "stream-K still on: %s; warnings: %d" % ("exact-fp32 (even F(4x4) schedule)" if F32 else "stream-K", B, N, dt * 1e3, bad, flags, [n.engine().stream_k is not False for n in nets], len(caught)))
assert bad == 0 and flags == [0, 0]
