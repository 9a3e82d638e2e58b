"""Micro-benchmark of the first layer (yv3_conv0) through the C-ABI; checks every mode against the fp32 output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu

B = int(os.environ.get("BB", "64")); S = int(os.environ.get("SIZE", "416")); iters = int(os.environ.get("ITERS", "20"))
torch.cuda.set_device(0)
lib = _ffi.lib()
m = conv_bn_relu(3, 32, 3, 1).cuda().eval()
with torch.no_grad():
    m.bn.running_var.uniform_(0.5, 1.5); m.bn.running_mean.uniform_(-0.2, 0.2); m.bn.weight.uniform_(0.5, 1.5); m.bn.bias.uniform_(-0.3, 0.3)
x = torch.rand(B, 3, S, S, device="cuda")
st = _ffi.stream_ptr()
ref = None
for name, dt in (("f32", _ffi.F32), ("f32h2", _ffi.F32H2), ("f32x3", _ffi.F32X3), ("bf16", _ffi.BF16)):
    pc = engine.pack_conv(m, m._spec(), dt)
    y = engine.alloc_act(B, S, S, 32, dt, "cuda")
    call = lambda: _ffi.check(lib.yv3_conv0(x.data_ptr(), pc.w.data_ptr(), pc.alpha.data_ptr(), pc.beta.data_ptr(), y.data_ptr(), B, S, S, dt, None, st))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    out = engine.from_planes(y, dt)
    if ref is None: ref = out
    err = ((out - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
    gb = (x.numel() * 4 + y.numel() * y.element_size()) / 1e9
    print("conv0 %-6s B=%d %dx%d: %.3f ms  %.2f TB/s algorithmic  max|d| vs f32 mode %.3g" % (name, B, S, S, ms, gb / ms, err)); sys.stdout.flush()
