import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/yolo_v3_amd") else ".")
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
torch.cuda.set_device(0)
lib = _ffi.lib()
B, cin, cout, H = 48, 512, 256, 52
m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
pc = engine.pack_conv(m, m._spec(), _ffi.F32X3)
x = engine.to_planes(torch.rand(B, H, H, cin, device="cuda") - 0.5, _ffi.F32X3)
y = engine.alloc_act(B, H, H, cout, _ffi.F32X3, "cuda")
d = engine.make_desc(pc, x, y, B, H, H, None, dtype=_ffi.F32X3)
for _ in range(3):
    _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
torch.cuda.synchronize()
a = pc.alpha.cpu()[:32].view(8, 4)
for w in range(8):
    wt, br, body, dma = a[w].tolist()
    nk = 144
    print("wave %d per chunk: vmcnt-wait %.0f  barrier %.0f  body %.0f  of which 9 DMA issues (incl. 2 memtime reads each) %.0f" % (w, wt / nk, br / nk, body / nk, dma / nk))
