"""F(4x4,3x3) launch time against the number of workgroups: is the launch a staircase in rounds of 2 x CUs workgroups?
   python tools/wino4_steps.py c26 40 41 42 44 48 56 64        (batch sizes; the GEMM stage's workgroups = ceil(T / 32) * cout / 64)
Prints per batch: workgroups, rounds of 512, ms of the launch pair (input transform + GEMM stage), ms per round-unit of work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
LAYERS = {"c26": (256, 512, 26), "c13": (512, 1024, 13), "c52": (128, 256, 52), "c104": (64, 128, 104)}
name = sys.argv[1]
batches = [int(a) for a in sys.argv[2:]]
iters = int(os.environ.get("ITERS", "10"))
cin, cout, H = LAYERS[name]
dt = _ffi.F32
torch.cuda.set_device(0)
lib = _ffi.lib(); st = _ffi.stream_ptr()
torch.manual_seed(cin + H)
m = conv_bn_relu(cin, cout, 3, 1).cuda().eval()
pc = engine.pack_conv(m, m._spec(), dt, winograd=True, winograd4=True)
Bmax = max(batches)
x = torch.rand(Bmax, H, H, cin, device="cuda") * 2 - 0.5
r = torch.rand(Bmax, H, H, cout, device="cuda") - 0.5
y = torch.empty(Bmax, H, H, cout, device="cuda")
ws = torch.zeros(lib.yv3_wino_workspace_bytes(Bmax, H, H, cin), dtype=torch.uint8, device="cuda")
descs = []
for B in batches:
    d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt, wino_ws=ws)
    d.options |= _ffi.OPT_WINO_ALWAYS
    descs.append(d)
    _ffi.check(lib.yv3_conv2d(d, st))
torch.cuda.synchronize()
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
for B, t in zip(batches, best):
    T = B * th * th
    wg = ((T + 31) // 32) * (cout // 64)
    fl = 2.0 * B * H * H * cout * cin * 9
    print("%-4s B=%3d  workgroups %5d = %.3f rounds of 512   %.4f ms  %.0f alg TF   %.4f ms per 512 workgroups" % (name, B, wg, wg / 512, t, fl / t / 1e9, t / (wg / 512)))
