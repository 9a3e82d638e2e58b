"""Micro-benchmark of single conv layers through the C-ABI (for kernel tuning)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, arch, engine
from yolo_v3_amd.darknet import conv_bn_relu

LAYERS = {  # name: cin, cout, k, stride, H (input), res
    "c52": (128, 256, 3, 1, 52, True), "c26": (256, 512, 3, 1, 26, True), "c13": (512, 1024, 3, 1, 13, True),
    "c104": (64, 128, 3, 1, 104, True), "c208": (32, 64, 3, 1, 208, True), "d208": (32, 64, 3, 2, 416, False),
    "p26": (512, 256, 1, 1, 26, False), "p52": (256, 128, 1, 1, 52, False), "p13": (1024, 512, 1, 1, 13, False),
    "p104": (128, 64, 1, 1, 104, False), "p208": (64, 32, 1, 1, 208, False),
    "g8k": (8192, 8192, 1, 1, 32, False),   # with BB=8: the 8192^3 GEMM of tools/gemm_ceiling_probe.py as a 1x1 layer (x3 MFMAs)
    "g4k3": (4096, 4096, 3, 1, 16, False),  # with BB=32: M=8192, N=4096, K=36864 through the 3x3 gather
    "L52": (512, 256, 3, 1, 52, True),      # long-K probe (K=4608): loop efficiency without prologue/epilogue weight
}
B = int(os.environ.get("BB", "64"))
iters = int(os.environ.get("ITERS", "10"))
dt = {"f32": _ffi.F32, "bf16": _ffi.BF16, "f32x3": _ffi.F32X3, "f32h2": _ffi.F32H2}[os.environ.get("DT", "f32")]
tdt = torch.float32
names = sys.argv[1:] or list(LAYERS)
torch.cuda.set_device(0)
lib = _ffi.lib()
for name in names:
    cin, cout, k, s, H, res = LAYERS[name]
    m = conv_bn_relu(cin, cout, k, s).cuda().eval()
    sp = m._spec()
    pc = engine.pack_conv(m, sp, dt)
    xf = torch.rand(B, H, H, cin, device="cuda") - 0.5
    if os.environ.get("ZERO") == "1":
        xf.zero_()
        with torch.no_grad():
            m.conv.weight.zero_()
    ho, wo = engine.out_hw(H, H, k, s)
    rf = (torch.rand(B, ho, wo, cout, device="cuda") - 0.5) if res else None
    x = engine.to_planes(xf, dt)
    r = engine.to_planes(rf, dt) if res else None
    y = engine.alloc_act(B, ho, wo, cout, dt, "cuda")
    ws = torch.zeros(lib.yv3_conv_workspace_bytes(), dtype=torch.uint8, device="cuda") if (dt == _ffi.F32H2 and os.environ.get("SK") == "1") else None
    d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt, workspace=ws)
    st = _ffi.stream_ptr()
    for _ in range(3):
        _ffi.check(lib.yv3_conv2d(d, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _ffi.check(lib.yv3_conv2d(d, st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * ho * wo * cout * cin * k * k
    extra = ""
    if dt != _ffi.F32 and os.environ.get("CHECK", "1") == "1":
        pc0 = engine.pack_conv(m, sp, _ffi.F32)
        y0 = torch.empty(B, ho, wo, cout, device="cuda")
        d0 = engine.make_desc(pc0, engine.from_planes(x, dt), y0, B, H, H, engine.from_planes(r, dt) if res else None, dtype=_ffi.F32)
        _ffi.check(lib.yv3_conv2d(d0, st)); torch.cuda.synchronize()
        err = ((engine.from_planes(y, dt) - y0).abs() / y0.abs().clamp(min=1.0)).max().item()
        extra = "  max|d| vs exact-fp32 kernel %.3g" % err
    print("%-5s B=%d %dx%d %d->%d k%d s%d : %.3f ms  %.1f TF" % (name, B, H, H, cin, cout, k, s, ms, fl / ms / 1e9) + extra); sys.stdout.flush()
