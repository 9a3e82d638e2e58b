"""Cycle split of one workgroup of the ping-pong conv kernel (needs a -DYV3_TIMELINE build: YV3_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
torch.cuda.set_device(0)
lib = _ffi.lib()
dt = _ffi.F32H2
for name, (cin, cout, H) in {"L52": (512, 256, 52), "c52": (128, 256, 52), "c104": (64, 128, 104)}.items():
    B = 64
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
    pc = engine.pack_conv(m, m._spec(), dt)
    x = engine.to_planes(torch.rand(B, H, H, cin, device="cuda") - 0.5, dt)
    r = engine.to_planes(torch.rand(B, H, H, cout, device="cuda") - 0.5, dt)
    y = engine.alloc_act(B, H, H, cout, dt, "cuda")
    d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt)
    for _ in range(3):
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
    torch.cuda.synchronize()
    a = pc.alpha.cpu()[:64].view(8, 8)
    sub = pc.alpha.cpu()[64:128].view(8, 8)
    nk = 9 * cin // 32
    print(name, "nk =", nk)
    for w in range(8):
        pro, load, b1, comp, b2, epi, drain, tot = a[w].tolist()
        print("  wave %d: prologue %6.0f | per chunk: load %5.0f  barrier %5.0f  compute %5.0f  barrier %5.0f | epilogue %6.0f  store drain %6.0f | total %7.0f cycles" % (w, pro, load / nk, b1 / nk, comp / nk, b2 / nk, epi, drain, tot))
        print("          load segment: ds_reads issued @%4.0f  dma_prepare done @%4.0f  vmcnt passed @%4.0f  (lgkmcnt(0) = end)" % tuple((sub[w, :3] / nk).tolist()))
