"""F(4x4,3x3) GEMM stage: one item per workgroup (tune[1] = 1) against the even schedule by the library's rule and with the number of parts
per tail item forced (tune[2] = 2 / 3 / 6), and the direct fp32 kernel (tune[0] = 10).  Same-box interleaved timing, difference of each schedule
from the one-item-per-workgroup result and from an fp64 convolution of two images, the kernel status word after the runs.
   python tools/wino4_even_ab.py c26:64 c26:42 c52:64 c52:30 c104:64 c13:64 c13:48          (VARIANTS=tiles,even,p2,p3,p6,direct)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
LAYERS = {"c26": (256, 512, 26), "c13": (512, 1024, 13), "c38": (256, 512, 38), "c19": (512, 1024, 19), "c52": (128, 256, 52), "c76": (128, 256, 76),
          "c104": (64, 128, 104), "c152": (64, 128, 152)}
iters = int(os.environ.get("ITERS", "10"))
dt = _ffi.F32
torch.cuda.set_device(0)
lib = _ffi.lib(); st = _ffi.stream_ptr()
for arg in sys.argv[1:]:
    name, B = arg.split(":"); B = int(B)
    cin, cout, H = LAYERS[name]
    torch.manual_seed(cin + H)
    m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
    with torch.no_grad():
        m.bn.weight.uniform_(0.6, 1.2); m.bn.bias.uniform_(-0.2, 0.2); m.bn.running_mean.uniform_(-0.2, 0.2); m.bn.running_var.uniform_(0.7, 1.4)
    pc = engine.pack_conv(m, m._spec(), dt, winograd=True, winograd4=True)
    x = torch.rand(B, H, H, cin, device="cuda") * 2 - 0.5
    r = torch.rand(B, H, H, cout, device="cuda") - 0.5
    ws = torch.zeros(lib.yv3_wino_workspace_bytes(B, H, H, cin), dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    CODES = {"tiles": (0, 1, 0), "even": (0, 0, 0), "p2": (0, 0, 2), "p3": (0, 0, 3), "p6": (0, 0, 6), "direct": (10, 0, 0)}
    variants = tuple((v, CODES[v]) for v in os.environ.get("VARIANTS", "tiles,even,p2,p3,p6,direct").split(","))
    ys, descs = [], []
    for _, code in variants:
        y = torch.full((B, H, H, cout), float("nan"), device="cuda")
        d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt, wino_ws=ws, flags=status)
        if code[0] == 0:
            d.options |= _ffi.OPT_WINO_ALWAYS
        d.tune[0], d.tune[1], d.tune[2] = code
        ys.append(y); descs.append(d)
        for _ in range(2):
            _ffi.check(lib.yv3_conv2d(d, st))
    torch.cuda.synchronize()
    nb = min(B, 2)
    with torch.no_grad():
        ref = F.conv2d(x[:nb].permute(0, 3, 1, 2).double(), m.conv.weight.double(), None, 1, 1)
        ref = F.batch_norm(ref, m.bn.running_mean.double(), m.bn.running_var.double(), m.bn.weight.double(), m.bn.bias.double(), False, 0.1, 1e-5)
        ref = F.leaky_relu(ref, 0.1) + r[:nb].permute(0, 3, 1, 2).double()
    errs = [float(((y[:nb].permute(0, 3, 1, 2).double() - ref).abs() / ref.abs().clamp(min=1.0)).max()) if torch.isfinite(y).all() else float("nan") for y in ys]
    dif = [float((y - ys[0]).abs().max()) for y in ys]
    best = [1e9] * len(descs)
    for rep in range(3):
        for i, d in enumerate(descs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                _ffi.check(lib.yv3_conv2d(d, st))
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters)
    th = (H + 3) // 4
    items = ((B * th * th + 31) // 32) * (cout // 64)
    print("%-4s B=%3d items %5d (%.3f rounds):" % (name, B, items, items / 512) +
          "".join("  %s %.4f ms err %.1e dif %.1e" % (v[0], t, e, df) for v, t, e, df in zip(variants, best, errs, dif)) +
          "  status %d flags_nonzero %d" % (int(status.item()), int((ws[-2600:] != 0).sum().item())))
    sys.stdout.flush()
