"""Determinism soak of the post-processing stage: the dense (608x608 bs=8 SW-dense), sparse and eval detections are post-processed
REPS times on two HIP streams at once (two PostProcessors, as the two lanes of a Detector do) and every result must equal the first
one bit for bit (boxes + counts).  The scan kernel hands keep words between the waves of a workgroup through LDS, the rank sort
scatters with atomics: a race would show up as a differing hash.   python tools/postproc_soak.py [REPS]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                    # noqa: E402
from yolo_v3_amd.utils import PostProcessor                     # noqa: E402
from tools.postproc_bench import CASES                          # noqa: E402


def digest(out, counts, B):
    h = counts.cpu()
    nk = h[B:].tolist()
    sha = hashlib.sha256(h.numpy().tobytes())
    o = out.cpu()
    for b in range(B):
        sha.update(o[b, :nk[b]].numpy().tobytes())
    return sha.hexdigest()[:16]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    bad = 0
    for name, (stream, size, B, seed, conf, nms, is_eval, max_cand) in CASES.items():
        net = bench.make_net(stream(), size, dev)
        with torch.no_grad():
            dets = net.forward_cat(bench.scenes(B, size, seed, dev)).clone()
        del net
        torch.cuda.empty_cache()
        half = B // 2
        parts = [dets[:half].contiguous(), dets[half:].contiguous()]
        N, C = dets.shape[1], dets.shape[2] - 5
        mc = max_cand or N
        pps = [PostProcessor(p.shape[0], N, C, dev, max_cand=mc, cap=mc) for p in parts]
        streams = [torch.cuda.Stream(device=dev) for _ in parts]
        want = None
        for it in range(reps):
            res = []
            for p, pp, st in zip(parts, pps, streams):
                with torch.cuda.stream(st):
                    res.append(pp.run_sync_free(p, conf, nms, is_eval, True, prob=True))
            torch.cuda.synchronize()
            got = tuple(digest(o, c, p.shape[0]) for (o, c), p in zip(res, parts))
            if want is None:
                want = got
            elif got != want:
                bad += 1
                print("%s: iteration %d differs: %s vs %s" % (name, it, got, want))
        print("%-6s %d x 2 concurrent post-processing runs, digest %s: %s" % (name, reps, want, "all identical" if not bad else "MISMATCHES"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
