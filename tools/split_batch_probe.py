"""13x13 3x3 layers at bs=64: 344 tiles of 256x128 = 1.34 rounds of the chip.  Does splitting the BATCH into a part that fills
exactly one round of big tiles and a remainder on 128x128 tiles (same bits: the K order does not depend on the tile) pay?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
torch.cuda.set_device(0)
lib = _ffi.lib(); dt = _ffi.F32H2
def make(cin, cout, H, B0, B1, tile1, res=True):
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval(); sp = m._spec(); pc = engine.pack_conv(m, sp, dt)
    B = B0 + B1
    xf = torch.rand(B, H, H, cin, device="cuda") - 0.5; rf = torch.rand(B, H, H, cout, device="cuda") - 0.5
    x, r, y = engine.to_planes(xf, dt), engine.to_planes(rf, dt), engine.alloc_act(B, H, H, cout, dt, "cuda")
    full = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt)
    parts = []
    for (b0, nb, tile) in ((0, B0, 0), (B0, B1, tile1)):
        if nb == 0: continue
        d = engine.make_desc(pc, x[:, b0:b0 + nb], y[:, b0:b0 + nb], nb, H, H, r[:, b0:b0 + nb], dtype=dt)
        # plane tensors are [2][B,...]: a batch slice keeps the plane stride of the FULL tensor -> not expressible in the desc
        parts.append(d)
    return full, parts, (x, r, y, pc, m)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
st = _ffi.stream_ptr()
for (cin, cout, H) in ((512, 1024, 13), (256, 512, 26)):
    for B in (64, 48, 32, 16):
        full, _, keep = make(cin, cout, H, B, 0, 0)
        ms = t(lambda: _ffi.check(lib.yv3_conv2d(full, st)))
        os.environ["YV3_TILE"] = "2"
        small, _, keep2 = make(cin, cout, H, B, 0, 0)
        os.environ["YV3_TILE"] = "0"
        ms2 = t(lambda: _ffi.check(lib.yv3_conv2d(small, st)))
        print("%d->%d @%d B=%2d: auto tile %.3f ms   128x128 tiles %.3f ms" % (cin, cout, H, B, ms, ms2))
