"""Per-layer conv timing from a rocprofv3 kernel trace CSV: maps the conv launches of the last
bench step to arch.conv_specs and prints achieved TFLOP/s per layer."""
import csv, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_v3_amd import arch
path, B, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
NAMES = ("conv_igemm", "conv_gemm1x1", "conv_wino4", "conv0_", "conv_planes", "conv_front", "conv_res64")
convs = []
pending = None                      # a Winograd layer = wino_input_kernel + the WINO conv_planes launch: one entry, durations added
for r in rows:
    kn = r["Kernel_Name"]
    if "wino_input_kernel" in kn or "wino_input_f32_kernel" in kn or "wino4_input_f32_kernel" in kn:       # (the input transform of the layer's form)
        pending = r
        continue
    if any(n in kn for n in NAMES):
        if pending is not None:
            r = dict(r)
            r["End_Timestamp"] = str(int(r["End_Timestamp"]) + int(pending["End_Timestamp"]) - int(pending["Start_Timestamp"]))
            r["Kernel_Name"] = kn.replace("conv_planes_kernel<", "winograd+conv_planes_kernel<")
            r["xform_ns"] = int(pending["End_Timestamp"]) - int(pending["Start_Timestamp"])
            pending = None
        convs.append(r)
fused_front = any("conv_front" in r["Kernel_Name"] for r in convs)      # feature.mlist.0 + .1 in one launch
fused_res = any("conv_res64" in r["Kernel_Name"] for r in convs)        # feature.mlist.2 (1x1 + 3x3 + add) in one launch
specs = arch.conv_specs(); hw = arch.conv_output_hw(size)
groups_of = ([[0, 1]] if fused_front else [[0], [1]]) + ([[2, 3]] if fused_res else [[2], [3]]) + [[i] for i in range(4, len(specs))]
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):              # bench.py's YV3_DUMP_PLAN: a layer may be two launches (batch split)
    import json
    pl = json.load(open(sys.argv[4]))
    groups_of = (([[0, 1]] + ([[2, 3]] if fused_res else [])) if fused_front else [[0]]) + [[i] for i in pl["desc_spec"][pl["first_desc"]:]]
# a 1x1 layer of the exact-fp32 mode may be TWO conv kernels (persistent GEMM + the rest on small tiles: yv3_conv2d_launches, dumped by bench.py):
# the pair becomes one entry, durations added
launches = None
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]) and "desc_launches" in pl:
    launches = [1] * (len(groups_of) - len(pl["desc_launches"][pl["first_desc"] * 0:])) + list(pl["desc_launches"])
per_step = len(groups_of)
merged, i = [], 0
while i < len(convs):
    r = convs[i]
    if launches and launches[len(merged) % per_step] == 2 and "conv_gemm1x1" in r["Kernel_Name"] and i + 1 < len(convs):
        r = dict(r)
        nx = convs[i + 1]
        r["End_Timestamp"] = str(int(r["End_Timestamp"]) + int(nx["End_Timestamp"]) - int(nx["Start_Timestamp"]))
        r["Kernel_Name"] = r["Kernel_Name"] + " + tiles"
        i += 1
    merged.append(r); i += 1
convs = merged
nsteps = len(convs) // per_step
last = convs[(nsteps - 1) * per_step: nsteps * per_step]
tot = 0.0; totf = 0.0
groups = {}
for r, idxs in zip(last, groups_of):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    fl = sum(2.0 * hw[i][0] * hw[i][1] * specs[i].cout * specs[i].cin * specs[i].k * specs[i].k * B for i in idxs)
    sp, (h, w) = specs[idxs[-1]], hw[idxs[-1]]
    kn = r["Kernel_Name"]
    cfg = ("conv_front (3->32 + 32->64 s2)" if "conv_front" in kn else "conv_res64 (64->32 1x1 + 32->64 3x3 + add)" if "conv_res64" in kn
           else "w4 192x128 four waves, 2 workgroups/CU " + kn[kn.find("<"):kn.find(">") + 1] if "conv_planes_w4" in kn
           else "F(4x4) 64x32 four waves, 2 workgroups/CU (+ input transform)" if "conv_wino4" in kn
           else "persistent GEMM 128x128 / 256x64, 8 waves, 1 workgroup/CU" + (" + rest on small tiles" if "+ tiles" in kn else "") if "conv_gemm1x1" in kn
           else kn[kn.find("<"):kn.find(">") + 1] + (" (+ input transform)" if "xform_ns" in r else "") if "<" in kn else "conv0")
    key = (sp.cin, sp.cout, sp.k, sp.stride, h, cfg)
    g = groups.setdefault(key, [0, 0.0, 0.0, 0.0]); g[0] += 1; g[1] += dur; g[2] += fl; g[3] += r.get("xform_ns", 0) * 1e-9 if isinstance(r, dict) else 0.0
    tot += dur; totf += fl
print("%-44s %3s %9s %8s %7s" % ("cin,cout,k,s,H,tile", "n", "ms total", "ms each", "TF"))
for key, (n, dur, fl, xf) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print("%-44s %3d %9.3f %8.3f %7.1f%s" % (str(key), n, dur * 1e3, dur * 1e3 / n, fl / dur / 1e12, "   (input transform %.3f ms each)" % (xf * 1e3 / n) if xf else ""))
print("total conv kernel time %.3f ms, %.1f TF" % (tot * 1e3, totf / tot / 1e12))
