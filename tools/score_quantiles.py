import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from yolo_v3_amd import synth
dev = torch.device("cuda:0")
for name, stream, size, B, seed, conf, ev in [("dense", synth.dense_weight_stream, 608, 8, 4, 0.5, False), ("eval", synth.eval_weight_stream, 416, 32, 5, 0.005, True), ("sparse", synth.weight_stream, 416, 64, 0, 0.5, False)]:
    net = bench.make_net(stream(), size, dev); x = bench.scenes(B, size, seed, dev)
    with torch.no_grad(): d = net.forward_cat(x)[0]
    sc = d[:, 5:] * d[:, 4:5]
    if ev:
        v = sc[sc > conf]
    else:
        best, cls = sc.max(1); m = best > conf; big = torch.bincount(cls[m]).argmax(); v = best[m & (cls == big)]
    q = torch.quantile(v.float().cpu(), torch.tensor([0, .01, .1, .25, .5, .75, .9, .99, 1.0]))
    print(name, v.numel(), [round(float(t), 5) for t in q])
    del net
