"""Same-process A/B of the network's first four convolutions (feature.mlist.0/1 and the first residual block):
fused kernels (conv_front.hip, conv_res64.hip) vs the launches they replace.  bs from BB (default 64), 416x416."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import YoloNet, WeightManager, synth, _ffi
torch.cuda.set_device(0)
B = int(os.environ.get("BB", "64"))
net = YoloNet((416, 416)).eval(); WeightManager(net).load_stream(synth.weight_stream()); net = net.cuda()
eng = net.engine(); eng.ensure_packed()
lib = _ffi.lib()
x = torch.from_numpy(synth.images(min(B, 16), 416, 1)).cuda().repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
plans = {}
for name, (ff, fr) in {"unfused": (False, False), "front": (True, False), "front+res64": (True, True)}.items():
    eng.fuse_front, eng.fuse_res64, eng._plans = ff, fr, {}
    plans[name] = eng.plan(B, 416, 416)
eng.fuse_front, eng.fuse_res64, eng._plans = True, True, {}
def run(plan):
    eng.run_front(plan, x)
    for j in range(plan.first_desc, 3):
        _ffi.check(lib.yv3_conv2d(ctypes.byref(plan.descs[j]), _ffi.stream_ptr()))
def t(plan, it=20):
    for _ in range(3): run(plan)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): run(plan)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for rnd in range(3):
    print("round %d: " % rnd + "   ".join("%s %.3f ms" % (n, t(p)) for n, p in plans.items()))
# the two fused kernels on their own
pf = plans["front+res64"]
def only_front():
    p0, p1, d1 = eng.packed[0], eng.packed[1], pf.descs[0]
    _ffi.check(lib.yv3_conv_front(x.data_ptr(), p0.w.data_ptr(), p0.alpha.data_ptr(), p0.beta.data_ptr(), p1.w.data_ptr(), p1.alpha.data_ptr(),
                                  p1.beta.data_ptr(), d1.y, B, 416, 416, pf.flags.data_ptr(), _ffi.stream_ptr()))
def only_res():
    p2, p3, d1, d3 = eng.packed[2], eng.packed[3], pf.descs[0], pf.descs[2]
    _ffi.check(lib.yv3_res_block64(d1.y, p2.w.data_ptr(), p2.alpha.data_ptr(), p2.beta.data_ptr(), p3.w.data_ptr(), p3.alpha.data_ptr(),
                                   p3.beta.data_ptr(), d3.y, B, 208, 208, pf.flags.data_ptr(), _ffi.stream_ptr()))
def tt(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("alone: conv_front %.3f ms   conv_res64 %.3f ms" % (tt(only_front), tt(only_res)))
outs = {n: (run(p), torch.cuda.synchronize(), p.layer_out["feature.mlist.2.conv2"].clone())[2] for n, p in plans.items()}
print("bit-identical:", all(torch.equal(outs["unfused"], o) for o in outs.values()))
