"""Host-side cost of one Detector call: enqueue time of run_device (no sync) and latency of the whole synchronous call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import synth, Detector
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
net = load_sw1_net(synth.weight_stream()).cuda()
for B in (1, 8, 32):
    x = torch.from_numpy(synth.images(B, 416, 5)).cuda()
    d = Detector(net, B, 416, 416)
    for _ in range(5): d(x)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): d.run_device(x)
    t_enq = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_async = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): d(x)
    t_sync = (time.perf_counter() - t0) / n
    print("B=%-2d lanes=%d: enqueue %.3f ms/call (host only), async throughput %.3f ms/call, synchronous call %.3f ms" % (B, d.lanes, t_enq * 1e3, t_async * 1e3, t_sync * 1e3)); sys.stdout.flush()

# enqueue cost without back-pressure from a full command queue: one step at a time, the GPU idle before each
for B in (1, 16, 64):
    x = torch.from_numpy(synth.images(B, 416, 5)).cuda()
    for lanes in (1, 2):
        if B == 1 and lanes == 2:
            continue
        d = Detector(net, B, 416, 416, lanes=lanes)
        for _ in range(5): d(x)
        ts = []
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d.run_device(x)
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        ts.sort()
        print("B=%-2d lanes=%d: enqueue of ONE step into an idle queue: median %.3f ms, min %.3f ms" % (B, lanes, ts[len(ts) // 2] * 1e3, ts[0] * 1e3)); sys.stdout.flush()
