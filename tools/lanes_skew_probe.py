"""Do two FREE-RUNNING, half-a-step SKEWED sub-batch pipelines beat the Detector's two lanes (which fork and join every step, so both lanes are always
in the same layer)?  Two independent one-lane Detectors (32 images each, own net copy = own plans) on two concurrent streams, results copied to the host
per pipeline, no cross-stream waits inside the loop:
   a  Detector(64, lanes=2) -- the shipped step (fork / join per step) + D2H          b  two pipelines, free-running, started together
   c  two pipelines, free-running, the second started `SKEW_MS` later (default: half a 32-image step)
All three process 64 images per iteration.   python tools/lanes_skew_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from yolo_v3_amd import Detector, synth
from yolo_v3_amd import detect as _d
import importlib
ydet = importlib.import_module("yolo_v3_amd.detect")

B, size = int(os.environ.get("BB", "64")), int(os.environ.get("SIZE", "416"))
N = int(os.environ.get("STEPS", "60"))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
stream = synth.weight_stream()
nets = [bench.make_net(stream, size, dev) for _ in range(3)]
x = bench.scenes(B, size, 1000, dev)
h = B // 2
pair = ydet.concurrent_stream_pair(dev, {})
assert pair is not None, "no concurrent stream pair on this GPU"
det2 = Detector(nets[0], B, size, size, lanes=2)
dA, dB = Detector(nets[1], h, size, size, lanes=1), Detector(nets[2], B - h, size, size, lanes=1)
host = torch.empty((B, 512, 7), dtype=torch.float32).pin_memory()
hc = torch.empty((2 * B,), dtype=torch.int32).pin_memory()
hostA, hostB = host[:h], host[h:]
hcA, hcB = torch.empty((2 * h,), dtype=torch.int32).pin_memory(), torch.empty((2 * (B - h),), dtype=torch.int32).pin_memory()
xa, xb = x[:h].contiguous(), x[h:].contiguous()


def shipped():
    boxes, counts = det2.run_device(x)
    host.copy_(boxes[:, :512], non_blocking=True)
    hc.copy_(counts, non_blocking=True)


def two_pipes():
    with torch.cuda.stream(pair[0]):
        b, c = dA.run_device(xa)
        hostA.copy_(b[:, :512], non_blocking=True); hcA.copy_(c, non_blocking=True)
    with torch.cuda.stream(pair[1]):
        b, c = dB.run_device(xb)
        hostB.copy_(b[:, :512], non_blocking=True); hcB.copy_(c, non_blocking=True)


def run(fn, skew_cycles=0):
    with torch.no_grad():
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
        if skew_cycles:
            with torch.cuda.stream(pair[1]):
                torch.cuda._sleep(skew_cycles)
        t0 = time.perf_counter()
        for _ in range(N):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


# the spin counter's rate: time a known spin
torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize()
cyc_per_ms = 20_000_000 / ((time.perf_counter() - t0) * 1e3)
for rep in range(3):
    ta = run(shipped)
    tb = run(two_pipes)
    half_step_ms = float(os.environ.get("SKEW_MS", "0")) or tb / 2
    tc = run(two_pipes, int(half_step_ms * cyc_per_ms))
    tc2 = run(two_pipes, int(0.25 * tb * cyc_per_ms))
    print("bs=%d %dx%d  shipped two lanes %.3f ms (%.0f img/s) | two free-running pipelines %.3f ms (%.0f) | + skew %.2f ms: %.3f ms (%.0f) | + skew %.2f ms: %.3f ms (%.0f)"
          % (B, size, size, ta, B / ta * 1e3, tb, B / tb * 1e3, half_step_ms, tc - half_step_ms / N, B / (tc - half_step_ms / N) * 1e3,
             0.25 * tb, tc2 - 0.25 * tb / N, B / (tc2 - 0.25 * tb / N) * 1e3))
    sys.stdout.flush()
