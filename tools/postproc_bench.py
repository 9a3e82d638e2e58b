"""Post-processing stages alone (filter / NMS) on detections the network produced, per workload:
    dense  608x608 bs=8  SW-dense weights (BASELINE configs[4]), conf 0.5 / nms 0.4
    sparse 416x416 bs=64 SW-1 weights (the headline), conf 0.5 / nms 0.4
    eval   416x416 bs=32 SW-eval weights, conf 0.005 / nms 0.45, is_eval
Prints us per stage (HIP events around `reps` back-to-back calls), the class histogram of the first image and a
checksum of the boxes (compare across builds: the post-processing is bit-exact).  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.   python tools/postproc_bench.py [dense sparse eval] [--sub N]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                    # noqa: E402
from yolo_v3_amd import synth                                   # noqa: E402
from yolo_v3_amd.utils import PostProcessor                     # noqa: E402

CASES = {"dense": (synth.dense_weight_stream, 608, 8, 4, 0.5, 0.4, False, None),
         "sparse": (synth.weight_stream, 416, 64, 0, 0.5, 0.4, False, None),
         "eval": (synth.eval_weight_stream, 416, 32, 5, 0.005, 0.45, True, 8192)}


def timed(fn, reps):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    names = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    sub = int(sys.argv[sys.argv.index("--sub") + 1]) if "--sub" in sys.argv else 0
    dev = torch.device("cuda:0")
    for name in names:
        stream, size, B, seed, conf, nms, is_eval, max_cand = CASES[name]
        net = bench.make_net(stream(), size, dev)
        x = bench.scenes(B, size, seed, dev)
        with torch.no_grad():
            dets = net.forward_cat(x).clone()
        del net
        torch.cuda.empty_cache()
        if sub:
            dets = dets[:sub].contiguous()
        Bq, N, A = dets.shape
        C = A - 5
        mc = max_cand or N
        pp = PostProcessor(Bq, N, C, dev, max_cand=mc, cap=mc)
        t_f = timed(lambda: pp.filter(dets, conf, is_eval, prob=True), 30)
        t_n = timed(lambda: pp.nms(dets, nms, True, mc, mc), 30)
        out, counts = pp.run_sync_free(dets, conf, nms, is_eval, True, prob=True)
        torch.cuda.synchronize()
        h = counts.cpu()
        ncand, nkeep = h[:Bq].tolist(), h[Bq:].tolist()
        sha = hashlib.sha256()
        for b in range(Bq):
            sha.update(out[b, :nkeep[b]].cpu().numpy().tobytes())
        # class histogram of image 0 from its kept+candidate keys is not exposed; derive it from the scores
        d0 = dets[0]
        sc = d0[:, 5:] * d0[:, 4:5]
        if is_eval:
            hist = (sc > conf).sum(0)
        else:
            best, cls = sc.max(1)
            hist = torch.bincount(cls[best > conf], minlength=C)
        hist = hist.cpu().tolist()
        n0 = sum(hist)
        print("%-6s B=%d N=%d  filter %7.1f us  nms %7.1f us  | cand/img %s  kept/img %s" %
              (name, Bq, N, t_f, t_n, ncand[:4], nkeep[:4]))
        print("       image 0: %d candidates in %d classes, largest %s, n^2 / sum n_c^2 = %.1f   boxes sha %s" %
              (n0, sum(1 for v in hist if v), sorted(hist)[-3:], n0 * n0 / max(1, sum(v * v for v in hist)), sha.hexdigest()[:16]))


if __name__ == "__main__":
    main()
