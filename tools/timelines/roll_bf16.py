"""Cycle split of one workgroup of the bf16 ROLLING tiles (conv_planes_kernel<1, 256|192, 256, 2, 4, 3, ..., ROLL> and the four-wave
<1, 256, 128, 2, 2, 3, ..., 2, 2, ROLL>) -- needs a -DYV3_TIMELINE build (YV3_MEASURE=1 YV3_LIB=...): prologue | per chunk: DMA address
preparation, first k-step block, waits, barrier, second k-step block | epilogue.   BB=16 python tools/timeline.py --kernel roll_bf16 c76 c38 [TILE code]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
torch.cuda.set_device(0)
lib = _ffi.lib()
dt = _ffi.BF16
B = int(os.environ.get("BB", "16"))
WG = int(os.environ.get("WG", "100"))
tile = int(os.environ.get("TILE", "0"))
LAYERS = {"c76": (128, 256, 76), "c38": (256, 512, 38), "c19": (512, 1024, 19), "c52": (128, 256, 52), "c26": (256, 512, 26), "c13": (512, 1024, 13)}
for name in (sys.argv[1:] or ["c76", "c38"]):
    cin, cout, H = LAYERS[name]
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
    pc = engine.pack_conv(m, m._spec(), dt)
    x = engine.to_planes(torch.rand(B, H, H, cin, device="cuda") - 0.5, dt)
    r = engine.to_planes(torch.rand(B, H, H, cout, device="cuda") - 0.5, dt)
    y = engine.alloc_act(B, H, H, cout, dt, "cuda")
    d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt)
    if tile:
        d.options = (d.options & ~(0xff << 8)) | (tile << 8)
    d.tune[2] = WG
    for _ in range(3):
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
    torch.cuda.synchronize()
    a = pc.alpha.cpu()[:80].view(8, 10)
    print(name, "B =", B, "tile code", tile, "workgroup", WG, "chunks", 9 * cin // 32)
    for w in range(8):
        prep, b0, wait, bar, b1, nk, pro, epi, tot, _ = a[w].tolist()
        if tot == 0:
            continue
        print("  wave %d: prologue %6.0f | per chunk: prepare %4.0f  block0 %5.0f  waits %5.0f  barrier %5.0f  block1 %5.0f (sum %5.0f) | epilogue %6.0f | total %7.0f cycles"
              % (w, pro, prep, b0, wait, bar, b1, prep + b0 + wait + bar + b1, epi, tot))
