"""Per-layer conv timing from a rocprofv3 kernel trace CSV: maps the conv launches of the last
bench step to arch.conv_specs and prints achieved TFLOP/s per layer."""
import csv, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_v3_amd import arch
path, B, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
convs = [r for r in rows if "conv_igemm" in r["Kernel_Name"] or "conv0_" in r["Kernel_Name"] or "conv_planes" in r["Kernel_Name"]
         or "conv_front" in r["Kernel_Name"]]
fused = any("conv_front" in r["Kernel_Name"] for r in convs)       # first two layers in one launch
per_step = 74 if fused else 75
nsteps = len(convs) // per_step
last = convs[(nsteps - 1) * per_step: nsteps * per_step]
specs = arch.conv_specs(); hw = arch.conv_output_hw(size)
if fused:                                                          # merge spec 0 into spec 1: FLOPs of both, shape of the second
    f0 = 2.0 * hw[0][0] * hw[0][1] * specs[0].cout * specs[0].cin * 9 * B
    specs, hw = specs[1:], hw[1:]
tot = 0.0; totf = 0.0
groups = {}
for r, sp, (h, w) in zip(last, specs, hw):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    fl = 2.0 * h * w * sp.cout * sp.cin * sp.k * sp.k * B
    kn = r["Kernel_Name"]; cfg = kn[kn.find("<"):kn.find(">") + 1] if "<" in kn else ("conv_front (3->32 + 32->64 s2)" if "conv_front" in kn else "conv0")
    if "conv_front" in kn:
        fl += f0
    key = (sp.cin, sp.cout, sp.k, sp.stride, h, cfg)
    g = groups.setdefault(key, [0, 0.0, 0.0]); g[0] += 1; g[1] += dur; g[2] += fl
    tot += dur; totf += fl
print("%-44s %3s %9s %8s %7s" % ("cin,cout,k,s,H,tile", "n", "ms total", "ms each", "TF"))
for key, (n, dur, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print("%-44s %3d %9.3f %8.3f %7.1f" % (str(key), n, dur * 1e3, dur * 1e3 / n, fl / dur / 1e12))
print("total conv kernel time %.3f ms, %.1f TF" % (tot * 1e3, totf / tot / 1e12))
