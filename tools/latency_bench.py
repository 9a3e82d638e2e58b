"""Small-batch latency of the fused pipeline (eager vs HIP-graph replay)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import synth, Detector, _ffi
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
net = load_sw1_net(synth.weight_stream()).cuda()
if os.environ.get("YV3_SK") in ("0", "1"):
    net.stream_k = os.environ["YV3_SK"] == "1"
modes = {"f32h2": _ffi.F32H2} if os.environ.get("ONLY_DEFAULT") else {"f32h2": _ffi.F32H2, "f32x3": _ffi.F32X3, "f32": _ffi.F32, "bf16": _ffi.BF16}
for B in [int(b) for b in (sys.argv[1:] or ["1", "4", "16"])]:
    x = torch.from_numpy(synth.images(B, 416, 7)).cuda()
    for name, mode in modes.items():
        for graph in (False, True):
            d = Detector(net, B, 416, 416, dtype=mode, graph=graph)
            for _ in range(5): d.run_device(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 30
            for _ in range(n): d.run_device(x)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            print("B=%d %-6s graph=%d : %.3f ms/batch  %.1f img/s" % (B, name, graph, ms, B / ms * 1e3)); sys.stdout.flush()
