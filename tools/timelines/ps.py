"""Cycle split of one persistent-stream workgroup (blockIdx 100) -- needs tools/probes/dead_experiments/conv_planes_persistent_stream.patch.txt applied
(the experiment of profiles/r04an_persistent_stream_ab.txt; not in the shipped kernels) and a -DYV3_TIMELINE build (YV3_LIB=...):
per wave: prologue (once) | per tile: first load segment (incl. the wait for the previous epilogue's stores), rest of the main loop, epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
torch.cuda.set_device(0)
lib = _ffi.lib()
dt = _ffi.F32H2
B = int(os.environ.get("BB", "64"))
for name, (cin, cout, H) in {"c52": (128, 256, 52), "c26": (256, 512, 26), "c104": (64, 128, 104)}.items():
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
    pc = engine.pack_conv(m, m._spec(), dt)
    x = engine.to_planes(torch.rand(B, H, H, cin, device="cuda") - 0.5, dt)
    r = engine.to_planes(torch.rand(B, H, H, cout, device="cuda") - 0.5, dt)
    y = engine.alloc_act(B, H, H, cout, dt, "cuda")
    d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt)
    d.tune[1] = 64
    for _ in range(3):
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
    torch.cuda.synchronize()
    a = pc.alpha.cpu()[:64].view(8, 8)
    print(name, "nk =", 9 * cin // 32)
    for w in range(8):
        pro, loop, first, _, _, epi, tiles, tot = a[w].tolist()
        print("  wave %d: prologue %6.0f | per tile: first segment %5.0f  loop %6.0f (%.0f per chunk)  epilogue %6.0f | tiles %d total %7.0f cycles (%.0f per tile)"
              % (w, pro, first, loop, loop / (9 * cin // 32), epi, tiles, tot, tot / max(tiles, 1)))
