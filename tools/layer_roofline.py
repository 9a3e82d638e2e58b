"""Per-layer roofline of the conv stack in the default fp16x2-plane mode: for every distinct layer shape of the last bench
step (tools/trace_layers.py output) the time the MFMA roof (2500/3 algorithmic TFLOP/s) and the HBM roof (6.3 TB/s achievable,
MI355X_MICROARCH.md) would need, against the measured time.  Algorithmic HBM bytes per launch: input once + residual once +
output once (4 B per element: two fp16 planes; fp32 for the head logits / the NCHW input) + packed weights once.
    python tools/layer_roofline.py profiles/r02b_layers.txt 64 > profiles/r02b_layer_roofline.txt"""
import ast
import sys

path, B = sys.argv[1], int(sys.argv[2])
PEAK_F, PEAK_B = 2500e12 / 3, 6.3e12
rows = []
for line in open(path):
    line = line.strip()
    if not line.startswith("("):
        continue
    key = ast.literal_eval(line[:line.rindex(")") + 1])
    n, tot, each, tf = line[line.rindex(")") + 1:].split()
    rows.append((key, int(n), float(each)))
print("%-34s %2s %8s %8s %8s %8s %6s %5s" % ("cin,cout,k,s,Hout", "n", "GFLOP", "MB", "t_mfma", "t_hbm", "meas", "frac"))
tot_meas = tot_roof = tot_mfma = 0.0
for (cin, cout, k, s, H, tile), n, ms in rows:
    M = B * H * H
    flops = 2.0 * M * cout * cin * k * k
    hin = H * s
    res = (k == 3 and s == 1 and cin * 2 == cout and cin >= 32 and tile != "conv0")        # res_layer.conv2 (also the branch 3x3s: see note)
    in_bytes = B * hin * hin * cin * (4 if cin != 3 else 4)
    if str(tile).startswith("conv_front"):                             # fused front: reads the 3-channel image, FLOPs of both layers
        in_bytes = B * hin * hin * 3 * 4
        flops += 2.0 * B * hin * hin * 32 * 27
    out_bytes = M * cout * 4
    w_bytes = cout * cin * k * k * 4
    if str(tile).startswith("conv_res64"):                             # fused residual block: x (64 ch) read once (input AND residual), 1x1 + 3x3 FLOPs
        in_bytes = M * 64 * 4
        flops += 2.0 * M * 32 * 64
        w_bytes += 32 * 64 * 4
        res = False
    byts = in_bytes + out_bytes + w_bytes + (out_bytes if res else 0)
    t_f, t_b = flops / PEAK_F * 1e3, byts / PEAK_B * 1e3
    roof = max(t_f, t_b)
    tot_meas += n * ms; tot_roof += n * roof; tot_mfma += n * t_f
    print("%-34s %2d %8.1f %8.1f %8.3f %8.3f %6.3f %5.2f %s" % (str((cin, cout, k, s, H)), n, flops / 1e9, byts / 1e6, t_f, t_b, ms, roof / ms,
                                                                 "HBM" if t_b > t_f else ""))
print("sum over the step: measured %.3f ms; MFMA-only roof %.3f ms (%.3f); per-layer max(MFMA, HBM) roof %.3f ms (%.3f)"
      % (tot_meas, tot_mfma, tot_mfma / tot_meas, tot_roof, tot_roof / tot_meas))
print("note: residual traffic is counted for every 3x3 stride-1 C/2->C layer; 18 of those (the branch convs) have none, which makes their HBM time an upper bound")
