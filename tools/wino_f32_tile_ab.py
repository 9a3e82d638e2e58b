"""Same-box A/B of the exact-fp32 Winograd stage's two tiles (csrc/conv_igemm_f32.hip launch_wino_f32): the eight-wave 128x128 tile (one
workgroup per CU; tune[0] = 9) vs the four-wave 64x128 tile (two per CU; tune[0] = 8) vs the shipped rule (tune[0] = 0); interleaved timing +
bitwise comparison (same K order per accumulator).   BB=64 python tools/wino_f32_tile_ab.py c26 c52 c13 c104"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
LAYERS = {"c26": (256, 512, 26), "c13": (512, 1024, 13), "c38": (256, 512, 38), "c19": (512, 1024, 19), "c52": (128, 256, 52), "c76": (128, 256, 76),
          "c104": (64, 128, 104), "c152": (64, 128, 152)}
B = int(os.environ.get("BB", "64"))
iters = int(os.environ.get("ITERS", "10"))
dt = _ffi.F32
torch.cuda.set_device(0)
lib = _ffi.lib(); st = _ffi.stream_ptr()
for name in sys.argv[1:] or ["c26", "c52", "c13", "c104"]:
    cin, cout, H = LAYERS[name]
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
    pc = engine.pack_conv(m, m._spec(), dt, winograd=True)
    x = torch.rand(B, H, H, cin, device="cuda") - 0.5
    r = torch.rand(B, H, H, cout, device="cuda") - 0.5
    ws = torch.zeros(lib.yv3_wino_workspace_bytes(B, H, H, cin), dtype=torch.uint8, device="cuda")
    variants = (("rule", 0), ("8 waves 128x128", 9), ("4 waves 64x128 x2/CU", 8))
    ys, descs = [], []
    for _, code in variants:
        y = torch.zeros(B, H, H, cout, device="cuda")
        d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt, wino_ws=ws)
        d.options |= _ffi.OPT_WINO_ALWAYS
        d.tune[0] = code
        ys.append(y); descs.append(d)
        for _ in range(2):
            _ffi.check(lib.yv3_conv2d(d, st))
    torch.cuda.synchronize()
    same = [bool(torch.equal(y, ys[1])) for y in ys]
    best = [1e9] * len(descs)
    for rep in range(3):
        for i, d in enumerate(descs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                _ffi.check(lib.yv3_conv2d(d, st))
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters)
    fl = 2.0 * B * H * H * cout * cin * 9
    t128 = -(-(B * ((H + 1) // 2) ** 2) // 128) * (cout // 128)
    print("%-4s B=%d %dx%d %d->%d (%d 128-row tiles):" % (name, B, H, H, cin, cout, t128) +
          "".join("  %s: %.4f ms %.0f alg TF%s" % (v[0], t, fl / t / 1e9, "" if ok else " (DIFFERS)") for v, t, ok in zip(variants, best, same)))
    sys.stdout.flush()
