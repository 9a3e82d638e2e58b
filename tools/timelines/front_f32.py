"""Barrier-wait vs total cycles of one consumer (matrix-core) and one producer (vector-ALU) wave of conv_front_f32_kernel's workgroup 17
(debug build with -DYV3_FRONT_TL; YV3_MEASURE=1 YV3_LIB points at it).  416x416, bs from BB (64).
Belongs to the producer / consumer version of the kernel kept as tools/probes/dead_experiments/conv_front_f32_specialised_waves.hip.txt (the shipped serial kernel
has no timeline marks): copy that file over csrc/conv_front_f32.hip, tools/build_variant.sh ftl "-DYV3_FRONT_TL", run this.  profiles/r05t_*."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolo_v3_amd import YoloNet, WeightManager, synth, _ffi
torch.cuda.set_device(0)
B = int(os.environ.get("BB", "64"))
net = YoloNet((416, 416)).eval(); WeightManager(net).load_stream(synth.weight_stream()); net = net.cuda()
eng = net.engine(_ffi.F32); eng.ensure_packed()
plan = eng.plan(B, 416, 416)
x = torch.rand(B, 3, 416, 416, device="cuda")
for _ in range(3):
    eng.run_front(plan, x)
torch.cuda.synchronize()
d = plan.layer_out["feature.mlist.1"].view(-1)[:16].cpu()
for nm, o in (("consumer wave 0", 0), ("producer wave 4", 8)):
    n = float(d[o + 2])
    print("%s: tiles %d | per tile: total %.0f cycles, of which waiting at the barrier %.0f (%.0f %%)" %
          (nm, n, float(d[o + 1]) / n, float(d[o]) / n, 100 * float(d[o]) / float(d[o + 1])))
    if o:
        print("   producer phases per tile (3 iterations): fetch issue %.0f | fma chain %.0f | BN + image write %.0f | input wait + copy (+ barrier) %.0f" %
              tuple(float(d[o + 3 + q]) / n for q in range(4)))
