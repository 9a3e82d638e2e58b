"""Plain 1x1 layers of the exact-fp32 mode: conv_igemm_f32's tiles (tune[0] = 13) against the persistent DMA-fed GEMM (csrc/conv_gemm_f32.hip,
tune[0] = 14) and the library's own choice.  Same-box interleaved timing, bit-equality of the two, error against an fp64 convolution.
   python tools/gemm1x1_ab.py p52:64 p26:64 p13:64 p104:64 p26:16"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
LAYERS = {"p52": (256, 128, 52), "p26": (512, 256, 26), "p13": (1024, 512, 13), "p104": (128, 64, 104), "p76": (256, 128, 76), "p38": (512, 256, 38),
          "p19": (1024, 512, 19),
          # 3x3 stride-2 layers (input side H): name -> (cin, cout, H, k, stride)
          "s104": (64, 128, 208, 3, 2), "s52": (128, 256, 104, 3, 2), "s26": (256, 512, 52, 3, 2), "s13": (512, 1024, 26, 3, 2),
          "s152": (64, 128, 304, 3, 2), "s76": (128, 256, 152, 3, 2), "s38": (256, 512, 76, 3, 2), "s19": (512, 1024, 38, 3, 2)}
iters = int(os.environ.get("ITERS", "10"))
dt = _ffi.F32
torch.cuda.set_device(0)
lib = _ffi.lib(); st = _ffi.stream_ptr()
for arg in sys.argv[1:]:
    name, B = arg.split(":"); B = int(B)
    cin, cout, H, k, stride = (LAYERS[name] + (1, 1))[:5]
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    torch.manual_seed(cin + H)
    m = conv_bn_relu(cin, cout, k, stride).cuda().eval()
    with torch.no_grad():
        m.bn.weight.uniform_(0.6, 1.2); m.bn.bias.uniform_(-0.2, 0.2); m.bn.running_mean.uniform_(-0.2, 0.2); m.bn.running_var.uniform_(0.7, 1.4)
    pc = engine.pack_conv(m, m._spec(), dt)
    x = torch.rand(B, H, H, cin, device="cuda") * 2 - 0.5
    variants = (("tiles", (13, 0)), ("gemm", (14, 0)), ("gemm_all", (14, 3)), ("auto", (0, 0)))
    ys, descs = [], []
    for _, code in variants:
        y = torch.full((B, Ho, Ho, cout), float("nan"), device="cuda")
        d = engine.make_desc(pc, x, y, B, H, H, None, dtype=dt)
        d.tune[0], d.tune[1] = code
        ys.append(y); descs.append(d)
        for _ in range(2):
            _ffi.check(lib.yv3_conv2d(d, st))
    torch.cuda.synchronize()
    nb = min(B, 2)
    with torch.no_grad():
        ref = F.conv2d(x[:nb].permute(0, 3, 1, 2).double(), m.conv.weight.double(), None, stride, k // 2)
        ref = F.batch_norm(ref, m.bn.running_mean.double(), m.bn.running_var.double(), m.bn.weight.double(), m.bn.bias.double(), False, 0.1, 1e-5)
        ref = F.leaky_relu(ref, 0.1)
    errs = [float(((y[:nb].permute(0, 3, 1, 2).double() - ref).abs() / ref.abs().clamp(min=1.0)).max()) if torch.isfinite(y).all() else float("nan") for y in ys]
    same = [bool(torch.equal(y, ys[0])) for y in ys]
    best = [1e9] * len(descs)
    for rep in range(3):
        for i, d in enumerate(descs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                _ffi.check(lib.yv3_conv2d(d, st))
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters)
    fl = 2.0 * B * Ho * Ho * cout * cin * k * k
    print("%-4s B=%3d %dx%d %d->%d:" % (name, B, H, H, cin, cout) +
          "".join("  %s %.4f ms %.0f TF err %.1e same %d" % (v[0], t, fl / t / 1e9, e, sm) for v, t, e, sm in zip(variants, best, errs, same)))
    if os.environ.get("TL"):            # measurement builds with -DG1_TIMELINE=<workgroup>: per wave [wait, barrier, burst] ticks per chunk, epilogue ticks per tile, total ticks, chunks, tiles
        _ffi.check(lib.yv3_conv2d(descs[2], st)); torch.cuda.synchronize()
        print("     timeline (s_memtime ticks): " + "  ".join("w%d %s" % (w, ["%.0f" % v for v in ys[2].view(-1)[w * 8:w * 8 + 7].tolist()]) for w in range(8)))
    sys.stdout.flush()
