"""Same-box A/B: Winograd F(2x2,3x3) form vs the direct fp16-plane kernel on single layers (C-ABI), interleaved timing.
   BB=64 python tools/wino_ab.py c26 c13"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
LAYERS = {"c26": (256, 512, 26), "c13": (512, 1024, 13), "c38": (256, 512, 38), "c19": (512, 1024, 19), "c52": (128, 256, 52), "c76": (128, 256, 76)}
B = int(os.environ.get("BB", "64"))
iters = int(os.environ.get("ITERS", "20"))
engine.WINO_MIN_CIN = 128
dt = _ffi.F32H2
torch.cuda.set_device(0)
lib = _ffi.lib(); st = _ffi.stream_ptr()
for name in sys.argv[1:] or ["c26", "c13"]:
    cin, cout, H = LAYERS[name]
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
    pc = engine.pack_conv(m, m._spec(), dt, winograd=True)
    xf = torch.rand(B, H, H, cin, device="cuda") - 0.5
    x = engine.to_planes(xf, dt)
    r = engine.to_planes(torch.rand(B, H, H, cout, device="cuda") - 0.5, dt)
    ws = torch.zeros(lib.yv3_wino_workspace_bytes(B, H, H, cin), dtype=torch.uint8, device="cuda")
    ys = [engine.alloc_act(B, H, H, cout, dt, "cuda") for _ in range(4)]
    descs = [engine.make_desc(pc, x, ys[0], B, H, H, r, dtype=dt), engine.make_desc(pc, x, ys[1], B, H, H, r, dtype=dt, wino_ws=ws),
             engine.make_desc(pc, x, ys[2], B, H, H, r, dtype=dt, wino_ws=ws), engine.make_desc(pc, x, ys[3], B, H, H, r, dtype=dt, wino_ws=ws)]
    descs[1].options |= _ffi.OPT_WINO_ALWAYS | _ffi.OPT_WINO_EVEN
    descs[2].options |= _ffi.OPT_WINO_ALWAYS
    descs[3].options |= _ffi.OPT_WINO_ALWAYS
    descs[3].tune[1] |= int(os.environ.get("ALT", "2"))   # ALT=2: the OTHER main loop of the GEMM stage (rolling <-> ping-pong, csrc/conv_planes.hip launch_wino)
    for d in descs:
        for _ in range(3):
            _ffi.check(lib.yv3_conv2d(d, st))
    torch.cuda.synchronize()
    a, b = engine.from_planes(ys[0], dt), engine.from_planes(ys[1], dt)
    err = ((a - b).abs() / a.abs().clamp(min=1.0)).max().item()
    same = bool(torch.equal(ys[2], ys[3]))
    best = [1e9, 1e9, 1e9, 1e9]
    for rep in range(3):
        for i, d in enumerate(descs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                _ffi.check(lib.yv3_conv2d(d, st))
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters)
    fl = 2.0 * B * H * H * cout * cin * 9
    print("%-4s B=%d %dx%d %d->%d : direct %.4f ms (%.0f alg TF)   winograd even schedule %.4f ms x%.2f   tile schedule %.4f ms (%.0f alg TF) x%.2f   rolling main loop %.4f ms (%.0f alg TF) x%.2f %s   max|d| %.3g"
          % (name, B, H, H, cin, cout, best[0], fl / best[0] / 1e9, best[1], best[0] / best[1], best[2], fl / best[2] / 1e9, best[0] / best[2],
             best[3], fl / best[3] / 1e9, best[0] / best[3], "bit-identical" if same else "DIFFERS", err)); sys.stdout.flush()
