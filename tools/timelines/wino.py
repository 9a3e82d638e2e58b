"""Cycle split of one workgroup (blockIdx 100) of the Winograd GEMM stage (ping-pong loop) -- needs a -DYV3_TIMELINE build
(YV3_LIB=...): per wave: prologue | per chunk: load segment, barrier, compute segment (+ fold), barrier | epilogue (4 outputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
torch.cuda.set_device(0)
lib = _ffi.lib()
dt = _ffi.F32H2
B = int(os.environ.get("BB", "64"))
for name, (cin, cout, H) in {"c13": (512, 1024, 13), "c26": (256, 512, 26)}.items():
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
    pc = engine.pack_conv(m, m._spec(), dt, winograd=True)
    x = engine.to_planes(torch.rand(B, H, H, cin, device="cuda") - 0.5, dt)
    r = engine.to_planes(torch.rand(B, H, H, cout, device="cuda") - 0.5, dt)
    y = engine.alloc_act(B, H, H, cout, dt, "cuda")
    ws = torch.zeros(lib.yv3_wino_workspace_bytes(B, H, H, cin), dtype=torch.uint8, device="cuda")
    d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt, wino_ws=ws)
    d.options |= _ffi.OPT_WINO_ALWAYS
    d.tune[1] |= int(os.environ.get("ALT", "0"))
    for _ in range(3):
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
    torch.cuda.synchronize()
    a = pc.alpha_wino.cpu()[:64].view(8, 8)
    print(name, "chunks per tile =", 16 * cin // 32)
    for w in range(8):
        pro, load, b1, comp, b2, epi, items, tot = a[w].tolist()
        print("  wave %d: prologue %6.0f | per chunk: load %5.0f  barrier %5.0f  compute %5.0f  barrier %5.0f (sum %5.0f) | epilogue %6.0f | total %7.0f cycles"
              % (w, pro, load, b1, comp, b2, load + b1 + comp + b2, epi, tot))
