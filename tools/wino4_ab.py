"""Same-box A/B of the exact-fp32 mode's three forms of a 3x3 / stride-1 layer: direct (tune[0] = 10, no OPT_WINO_ALWAYS), Winograd F(2x2,3x3)
(filters packed without the F(4x4) image) and F(4x4,3x3) (csrc/conv_wino4_f32.hip).  Interleaved timing, error of each form against an fp64
convolution of a few images, and the input transform's share.   BB=64 python tools/wino4_ab.py c52 c26 c13 c104"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
LAYERS = {"c26": (256, 512, 26), "c13": (512, 1024, 13), "c38": (256, 512, 38), "c19": (512, 1024, 19), "c52": (128, 256, 52), "c76": (128, 256, 76),
          "c104": (64, 128, 104), "c152": (64, 128, 152)}
B = int(os.environ.get("BB", "64"))
iters = int(os.environ.get("ITERS", "10"))
dt = _ffi.F32
torch.cuda.set_device(0)
lib = _ffi.lib(); st = _ffi.stream_ptr()
for name in sys.argv[1:] or ["c52", "c26", "c13", "c104"]:
    cin, cout, H = LAYERS[name]
    torch.manual_seed(cin + H)
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
    with torch.no_grad():
        m.bn.weight.uniform_(0.6, 1.2); m.bn.bias.uniform_(-0.2, 0.2); m.bn.running_mean.uniform_(-0.2, 0.2); m.bn.running_var.uniform_(0.7, 1.4)
    pc4 = engine.pack_conv(m, m._spec(), dt, winograd=True, winograd4=True)
    pc2 = engine.pack_conv(m, m._spec(), dt, winograd=True, winograd4=False)
    assert pc4.w_wino4 is not None and pc2.w_wino4 is None
    x = torch.rand(B, H, H, cin, device="cuda") * 2 - 0.5
    r = None if os.environ.get("NORES") else torch.rand(B, H, H, cout, device="cuda") - 0.5
    ws = torch.zeros(lib.yv3_wino_workspace_bytes(B, H, H, cin), dtype=torch.uint8, device="cuda")
    variants = (("direct", pc2, 0, 10), ("F(2x2)", pc2, _ffi.OPT_WINO_ALWAYS, 0), ("F(4x4)", pc4, _ffi.OPT_WINO_ALWAYS, 0))
    ys, descs = [], []
    for _, pc, opt, code in variants:
        y = torch.full((B, H, H, cout), float("nan"), device="cuda")
        d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt, wino_ws=(ws if opt else None))
        d.options |= opt
        d.tune[0] = code
        ys.append(y); descs.append(d)
        for _ in range(2):
            _ffi.check(lib.yv3_conv2d(d, st))
    forms = [lib.yv3_conv2d_form(d) for d in descs]
    torch.cuda.synchronize()
    nb = min(B, 2)
    with torch.no_grad():
        xr = x[:nb].permute(0, 3, 1, 2).double()
        ref = F.conv2d(xr, m.conv.weight.double(), None, 1, 1)
        ref = F.batch_norm(ref, m.bn.running_mean.double(), m.bn.running_var.double(), m.bn.weight.double(), m.bn.bias.double(), False, 0.1, 1e-5)
        ref = F.leaky_relu(ref, 0.1) + (r[:nb].permute(0, 3, 1, 2).double() if r is not None else 0)
    errs = []
    for y in ys:
        g = y[:nb].permute(0, 3, 1, 2).double()
        errs.append(float(((g - ref).abs() / ref.abs().clamp(min=1.0)).max()) if torch.isfinite(y).all() else float("nan"))
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
    print("%-4s B=%d %dx%d %d->%d:" % (name, B, H, H, cin, cout) +
          "".join("  %s (form %d): %.4f ms %.0f alg TF, err %.2e" % (v[0], f, t, fl / t / 1e9, e) for v, t, f, e in zip(variants, best, forms, errs)))
    if os.environ.get("TL"):            # measurement builds with -DW4F_TIMELINE=<workgroup>: per wave [sync, burst, fold] ticks per chunk, loop ticks, epilogue ticks, chunks
        _ffi.check(lib.yv3_conv2d(descs[2], st)); torch.cuda.synchronize()
        print("     timeline (s_memtime ticks, 100 MHz): " + "  ".join("w%d %s" % (w, ["%.1f" % v for v in ys[2].view(-1)[w * 8:w * 8 + 6].tolist()]) for w in range(4)))
    sys.stdout.flush()
