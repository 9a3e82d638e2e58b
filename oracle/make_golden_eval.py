"""Golden vectors for eval mode AS THE REFERENCE RUNS IT (evaluate.py:201-204: obj_conf_thr 0.005, nms_thr 0.45,
is_eval=True, use_nms=True), produced by the reference itself with the "SW-eval" synthetic weights
(yolo_v3_amd.synth.eval_weight_stream: ~1-2 k (row, class) candidates per image at 0.005).

    python oracle/make_golden_eval.py        # build container only (needs /root/reference)
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
GOLD = os.path.join(REPO, "tests", "golden")


def main():
    from make_golden import import_reference
    torch, darknet, yololayer, utils, boundingbox = import_reference()
    from yolo_v3_amd import synth
    torch.set_num_threads(8)
    wpath = "/tmp/sweval.weights"
    synth.write_darknet_weights(wpath, synth.eval_weight_stream(), seen=0)
    net = darknet.YoloNet((416, 416)).eval()
    darknet.WeightManager(net).loadWeight(wpath)
    g = {"cfg": np.array([0.005, 0.45])}
    B, size, seed = 2, 416, 3011
    x = torch.from_numpy(synth.images(B, size, seed))
    with torch.no_grad():
        d1, d2, d3 = net(x, None)                                                     # evaluate.py:201
        dets = torch.cat((d1, d2, d3), 1)
        res = utils.postprocessing(dets.clone(), 80, obj_conf_thr=0.005, nms_thr=0.45, is_eval=True, use_nms=True)   # :202-204
    sc = dets[..., 5:] * dets[..., 4:5]
    g["in_cfg"] = np.array([B, size, seed], dtype=np.int64)
    g["n_pairs"] = (sc > 0.005).sum((1, 2)).numpy()
    rows = np.arange(0, dets.shape[1], 37, dtype=np.int32)
    g["rows"], g["dets_rows"] = rows, dets[:, rows].numpy()
    for i, r in enumerate(res):
        g["boxes%d" % i] = r.numpy().astype(np.float32)
        print("image", i, "pairs", int(g["n_pairs"][i]), "kept", tuple(r.shape))
    np.savez_compressed(os.path.join(GOLD, "e2e_eval.npz"), **g)


if __name__ == "__main__":
    main()
