"""Cycle split of the fused front kernel (debug build with -DYV3_FRONT_TL; YV3_LIB points at it).  DT=f32h2|bf16, SIZE (416), BB (64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolo_v3_amd import YoloNet, WeightManager, synth, _ffi
torch.cuda.set_device(0)
S, B = int(os.environ.get("SIZE", "416")), int(os.environ.get("BB", "64"))
mode = {"f32h2": _ffi.F32H2, "bf16": _ffi.BF16}[os.environ.get("DT", "f32h2")]
net = YoloNet((S, S)).eval(); WeightManager(net).load_stream(synth.weight_stream()); net = net.cuda()
eng = net.engine(mode); eng.ensure_packed()
plan = eng.plan(B, S, S)
x = torch.rand(B, 3, S, S, device="cuda")
for _ in range(3):
    eng.run_front(plan, x)
torch.cuda.synchronize()
y = plan.layer_out["feature.mlist.1"]
d = y.view(-1)[:8 * 8 * 2].view(torch.float32).cpu().view(8, 8)
names = ["patch->LDS", "barrier waits", "first layer", "second conv", "epilogue"]
for w in range(8):
    n = float(d[w, 5])
    print("wave %d: tiles %d | " % (w, n) + "  ".join("%s %.0f" % (nm, float(d[w, i]) / n) for i, nm in enumerate(names)) + "  | total/tile %.0f" % (float(d[w, :5].sum()) / n))

# ---- conv_res64 (debug build with -DYV3_RES_TL)
if os.environ.get("RES_TL"):
    eng.run_front(plan, x); torch.cuda.synchronize()
    y = plan.layer_out["feature.mlist.2.conv2"]
    d = y.view(-1)[:8 * 8 * 2].view(torch.float32).cpu().view(8, 8)
    names = ["loop top", "barrier waits", "1x1 -> image", "3x3 conv", "epilogue", "vmcnt wait"]
    for w in range(8):
        n = float(d[w, 6])
        print("res64 wave %d: tiles %d | " % (w, n) + "  ".join("%s %.0f" % (nm, float(d[w, i]) / n) for i, nm in enumerate(names)) + "  | total/tile %.0f" % (float(d[w, :6].sum()) / n))
