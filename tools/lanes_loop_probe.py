"""Why does a two-lane step cost more inside bench.py's loop than inside the lane calibration?  Same Detector, loop variants:
   a  _run_lanes back to back (the calibration's loop)      b  run_device            c  b + the bench's two D2H copies
   d  c + stage marks (HIP events)                            e  d on the synthetic scenes instead of uniform noise
   BB=16 SIZE=416 python tools/lanes_loop_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from yolo_v3_amd import Detector, synth

B, size = int(os.environ.get("BB", "16")), int(os.environ.get("SIZE", "416"))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net = bench.make_net(synth.weight_stream(), size, dev)
gen = torch.Generator(device=dev).manual_seed(1234)
noise = torch.rand((B, 3, size, size), device=dev, generator=gen)
scene = bench.scenes(B, size, 1, dev)
noop = lambda name: None
for lanes in (1, 2, 1, 2):
    det = Detector(net, B, size, size, lanes=lanes)
    host = torch.empty((B, 512, 7), dtype=torch.float32).pin_memory()
    hc = torch.empty((2 * B,), dtype=torch.int32).pin_memory()

    def loop(fn, n=60):
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, t_host / n * 1e3

    def marks():
        def mark(name):
            ev = torch.cuda.Event(enable_timing=True); ev.record()
        return mark

    def c(x, mk=None):
        boxes, counts = det.run_device(x, mk)
        host.copy_(boxes[:, :512], non_blocking=True)
        hc.copy_(counts, non_blocking=True)

    res = {"a": loop(lambda: det._run_lanes(noise, noop)), "b": loop(lambda: det.run_device(noise)), "c": loop(lambda: c(noise)),
           "d": loop(lambda: c(noise, marks())), "e": loop(lambda: c(scene, marks()))}
    print("bs=%d %dx%d lanes=%d: " % (B, size, size, lanes) + "  ".join("%s %.3f ms (host %.3f)" % (k, v[0], v[1]) for k, v in res.items()))
    sys.stdout.flush()
    del det
