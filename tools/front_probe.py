"""Do the HBM-bound first layers run faster when their activations stay in the 256 MB Infinity Cache?
Times conv0 + the first NL descriptors of the plan for sub-batches of different sizes (per-image time)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import YoloNet, WeightManager, synth, _ffi
torch.cuda.set_device(0)
stream = synth.weight_stream()
net = YoloNet((416, 416)).eval(); WeightManager(net).load_stream(stream); net = net.cuda()
eng = net.engine(); eng.ensure_packed()
lib = _ffi.lib()
NL = int(os.environ.get("NL", "3"))
for B in (64, 32, 16, 8, 4):
    plan = eng.plan(B, 416, 416)
    x = torch.rand(B, 3, 416, 416, device="cuda")
    reps = 64 // B
    def run():
        for _ in range(reps):
            eng.run_conv0(plan, x)
            for j in range(NL):
                _ffi.check(lib.yv3_conv2d(ctypes.byref(plan.descs[j]), _ffi.stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print("sub-batch %2d x %d: conv0 + first %d convs for 64 images: %.3f ms" % (B, reps, NL, e0.elapsed_time(e1) / 10))
