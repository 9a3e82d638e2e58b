"""Head conv (1x1, cout = 255, fp32 output with the fused YOLO decode) against a plane-output 1x1 conv of the same GEMM shape, and
its epilogue IO ablations (tune[3]: bit 0 no stores, bit 2 no decode math -- results INVALID).
  DT=bf16 BB=16 python tools/head_probe.py 76 256      # grid, cin"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine, arch
from yolo_v3_amd.darknet import conv_bn_relu

B = int(os.environ.get("BB", "16"))
dt = {"bf16": _ffi.BF16, "f32h2": _ffi.F32H2}[os.environ.get("DT", "bf16")]
G, cin = int(sys.argv[1]), int(sys.argv[2])
torch.cuda.set_device(0)
lib, st = _ffi.lib(), _ffi.stream_ptr()


def timed(d, iters=30):
    for _ in range(3):
        _ffi.check(lib.yv3_conv2d(d, st))
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            _ffi.check(lib.yv3_conv2d(d, st))
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


x = engine.to_planes(torch.rand(B, G, G, cin, device="cuda") - 0.5, dt)
fl = 2.0 * B * G * G * cin * 255
# plane-output conv of the same shape
m = conv_bn_relu(cin, 256, 1, 1).cuda().eval()
pc = engine.pack_conv(m, m._spec(), dt)
y = engine.alloc_act(B, G, G, 256, dt, "cuda")
t = timed(engine.make_desc(pc, x, y, B, G, G, None, dtype=dt))
print("B=%d %dx%d %d->256 1x1 plane output           : %.4f ms  %.0f TF" % (B, G, G, cin, t, fl / t / 1e9))
# head conv
h = torch.nn.Conv2d(cin, 255, 1, 1, 0, bias=True).cuda().eval()
pch = engine.pack_conv(h, arch.ConvSpec("head", cin, 255, 1, 1, False, False), dt)
logits = torch.empty(B, G, G, 255, device="cuda")
dets = torch.empty(B, 3 * G * G, 85, device="cuda")
for label, use_dec, use_y, tune3 in (("logits only (fp32 [B,G,G,255])", False, True, 0), ("fused decode", True, False, 0),
                                     ("fused decode, no decode math", True, False, 4), ("fused decode, no stores", True, False, 1),
                                     ("fused decode, neither", True, False, 5)):
    d = engine.make_desc(pch, x, logits if use_y else None, B, G, G, None, dtype=dt, out_dtype=_ffi.F32)
    if use_dec:
        d.dec_out = dets.data_ptr(); d.dec_stride = 8.0; d.dec_out_batch_stride = 3 * G * G * 85
        for k, a in enumerate((10, 13, 16, 30, 33, 23)):
            d.dec_anchors[k] = float(a)
    d.tune[3] = tune3
    t = timed(d)
    print("B=%d %dx%d %d->255 head, %-32s: %.4f ms  %.0f TF" % (B, G, G, cin, label, t, fl / t / 1e9))
