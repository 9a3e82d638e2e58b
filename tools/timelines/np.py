"""Cycle split of one workgroup (blockIdx 17) of the SINGLE-PHASE (non ping-pong) plane-conv main loop -- needs a -DYV3_TIMELINE
build (YV3_LIB=...).  Per wave, over the whole K loop: vmcnt wait | barrier | body (MFMA units + LDS reads + DMA issue), and the
cycles spent issuing DMA pieces.   DT=bf16 YV3_TILE=5 BB=34 python tools/timeline.py --kernel np"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
torch.cuda.set_device(0)
lib = _ffi.lib()
dt = {"f32h2": _ffi.F32H2, "bf16": _ffi.BF16}[os.environ.get("DT", "bf16")]
B = int(os.environ.get("BB", "34"))
for name, (cin, cout, H) in {"c52": (128, 256, 52), "c26": (256, 512, 26), "c104": (64, 128, 104)}.items():
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
    pc = engine.pack_conv(m, m._spec(), dt)
    x = engine.to_planes(torch.rand(B, H, H, cin, device="cuda") - 0.5, dt)
    r = engine.to_planes(torch.rand(B, H, H, cout, device="cuda") - 0.5, dt)
    y = engine.alloc_act(B, H, H, cout, dt, "cuda")
    d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt)
    for _ in range(3):
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
    torch.cuda.synchronize()
    nk = 9 * cin // 32
    a = pc.alpha.cpu()[:32].view(8, 4)
    print(name, "nk =", nk, "(per chunk, cycles)")
    for w in range(8):
        wait, bar, body, dma = (a[w] / nk).tolist()
        if wait + bar + body > 0:
            print("  wave %d: vmcnt wait %5.0f  barrier %5.0f  body %5.0f (of which DMA issue %5.0f)  sum %5.0f" % (w, wait, bar, body, dma, wait + bar + body))
