"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's ``darknet`` / ``yololayer`` / ``utils`` modules with three
shims: stub modules for the unused ``cv2`` / ``torchvision`` imports,
``Tensor.cuda`` neutralised because the inference branch hard-codes ``.cuda()``
(yololayer.py:98-100), and a warnings filter.  Inputs come from ``yolo_v3_amd.synth``
(integer-RNG, regenerated bit-identically by the tests), so fixtures store outputs only,
plus the few hand-built inputs.

    python oracle/make_golden.py            # rewrites tests/golden/*
"""
import hashlib
import json
import os
import sys
import types
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def import_reference():
    sys.path.insert(0, "/root/reference")
    for n in ["cv2", "torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.models"]:
        sys.modules[n] = types.ModuleType(n)
    tv = sys.modules["torchvision"]
    tv.transforms, tv.datasets, tv.models = (sys.modules["torchvision." + k] for k in ("transforms", "datasets", "models"))
    import torch
    warnings.simplefilter("ignore")
    torch.Tensor.cuda = lambda self, *a, **k: self
    import darknet, yololayer, utils, boundingbox  # noqa: E401  (the reference's modules)
    return torch, darknet, yololayer, utils, boundingbox


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def letterbox_dog(size=416):
    """Input preparation for config 1 (own code; cv2 is absent so the reference's
    utils.letterbox_image cannot run).  602x452 -> 416x312 bicubic, centred on 128-grey."""
    from PIL import Image
    img = Image.open("/root/reference/imgs/dog-cycle-car.png").convert("RGB")
    w, h = img.size
    ratio = min(size / w, size / h)
    nw, nh = int(w * ratio), int(h * ratio)
    res = img.resize((nw, nh), Image.BICUBIC)
    canvas = np.full((size, size, 3), 128, dtype=np.uint8)
    y0, x0 = (size - nh) // 2, (size - nw) // 2
    canvas[y0:y0 + nh, x0:x0 + nw] = np.asarray(res)
    return canvas


def margins(torch, utils, dets, conf_thr, nms_thr, nc=80):
    """Distance of the decision inputs from their thresholds (for choosing robust fixtures)."""
    d = dets.clone()
    x1 = d[..., 0] - d[..., 2] / 2; x2 = d[..., 0] + d[..., 2] / 2
    y1 = d[..., 1] - d[..., 3] / 2; y2 = d[..., 1] + d[..., 3] / 2
    sc = d[..., 5:5 + nc] * d[..., 4:5]
    mx, arg = sc.max(-1)
    conf_margin = float((mx - conf_thr).abs().min())
    # margin between best and second-best class for rows that pass (argmax stability)
    top2 = sc.topk(2, -1)[0]
    passed = mx > conf_thr
    cls_margin = float((top2[..., 0] - top2[..., 1])[passed].min()) if passed.any() else 1.0
    iou_margin, tie_margin = 1.0, 1.0
    for b in range(d.shape[0]):
        idx = passed[b].nonzero().squeeze(1)
        if len(idx) < 2:
            continue
        boxes = torch.stack((x1[b, idx], y1[b, idx], x2[b, idx], y2[b, idx]), 1)
        iou = utils.iou_vectorized(boxes)
        same = arg[b, idx][:, None] == arg[b, idx][None, :]
        off = ~torch.eye(len(idx), dtype=torch.bool)
        sel = same & off
        if sel.any():
            iou_margin = min(iou_margin, float((iou[sel] - nms_thr).abs().min()))
            s = mx[b, idx]
            ds = (s[:, None] - s[None, :]).abs()
            tie_margin = min(tie_margin, float(ds[sel].min()))
    return dict(conf=conf_margin, cls=cls_margin, iou=iou_margin, tie=tie_margin, n_pass=int(passed.sum()))


def main():
    torch, darknet, yololayer, utils, boundingbox = import_reference()
    from yolo_v3_amd import synth
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    meta = {"torch": torch.__version__}

    # ---------------------------------------------------------------- G1: .weights round trip
    stream = synth.weight_stream()
    wpath = "/tmp/sw1.weights"
    synth.write_darknet_weights(wpath, stream, seen=32013312)
    net = darknet.YoloNet((416, 416)).eval()
    wm = darknet.WeightManager(net)
    ptr = wm.loadWeight(wpath)
    sd = net.state_dict()
    g1 = {"ptr": int(ptr), "header": [int(v) for v in wm.header], "seen": int(wm.seen),
          "n_convs": len(wm.conv_list), "keys": list(sd.keys()),
          "shapes": {k: list(v.shape) for k, v in sd.items()},
          "sha256": {k: sha(v.numpy()) for k, v in sd.items() if "num_batches" not in k},
          "stream_sha256": sha(stream)}
    # backbone-only loader (darknet.py:102-104)
    net_b = darknet.YoloNet((416, 416)).eval()
    g1["backbone_ptr"] = int(darknet.WeightManager(net_b.feature).loadWeight(wpath))
    json.dump(g1, open(os.path.join(GOLD, "weights_roundtrip.json"), "w"), indent=0)
    print("G1 ptr", ptr, "backbone", g1["backbone_ptr"])

    # ---------------------------------------------------------------- G2: decode
    anchors = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]
    g2 = {}
    for (h, size, mask, step, seed) in [(13, 416, [6, 7, 8], 1, 101), (26, 416, [3, 4, 5], 3, 102),
                                        (52, 416, [0, 1, 2], 11, 103), (19, 608, [6, 7, 8], 1, 104),
                                        (76, 608, [0, 1, 2], 29, 105)]:
        B = 2
        logits = synth.uniform(seed, 7, B * 255 * h * h, -6.0, 6.0).reshape(B, 255, h, h)
        layer = yololayer.YoloLayer(anchors, mask, (size, size), 80)
        with torch.no_grad():
            out = layer(torch.from_numpy(logits.copy()), (size, size), None).numpy()
        key = "h%d_s%d" % (h, size)
        g2[key + "_rows"] = np.arange(0, out.shape[1], step, dtype=np.int32)
        g2[key + "_out"] = out[:, ::step].copy()
        g2[key + "_sha"] = np.frombuffer(bytes.fromhex(sha(out)), dtype=np.uint8)
        g2[key + "_cfg"] = np.array([h, size, step, seed] + mask, dtype=np.int32)
    np.savez_compressed(os.path.join(GOLD, "decode.npz"), **g2)
    print("G2 done")

    # ---------------------------------------------------------------- G4: IOU
    raw = synth.uniform(201, 1, 64 * 4, 0.0, 200.0).reshape(64, 4)
    xyxy = np.stack((raw[:, 0], raw[:, 1], raw[:, 0] + raw[:, 2] * 0.5 + 1, raw[:, 1] + raw[:, 3] * 0.5 + 1), 1).astype(np.float32)
    xyxy[5] = xyxy[4]                       # identical pair
    xyxy[9, 2] = xyxy[9, 0]                 # zero-area box -> NaN self IOU
    b2 = xyxy[::-1][:40].copy()
    cxcywh = np.stack(((xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2, xyxy[:, 2] - xyxy[:, 0], xyxy[:, 3] - xyxy[:, 1]), 1).astype(np.float32)
    g4 = dict(xyxy=xyxy, b2=b2, cxcywh=cxcywh,
              iou_vec=utils.iou_vectorized(torch.from_numpy(xyxy)).numpy(),
              bbox_iou_xyxy=utils.bbox_iou(torch.from_numpy(xyxy), torch.from_numpy(b2)).numpy(),
              bbox_iou_cxcywh=utils.bbox_iou(torch.from_numpy(cxcywh), torch.from_numpy(cxcywh[:40].copy()), mode="cxcywh").numpy(),
              to_xyxy=boundingbox.bbox_cxcywh_to_x1y1x2y2(torch.from_numpy(cxcywh.copy())).numpy())
    np.savez_compressed(os.path.join(GOLD, "iou.npz"), **g4)
    print("G4 done")

    # ---------------------------------------------------------------- G3: post-processing cases
    def mk(B, N, nc=80):
        d = np.zeros((B, N, 5 + nc), dtype=np.float32)
        d[..., 2:4] = 10.0          # non-degenerate default boxes far apart
        d[..., 0] = np.arange(N)[None, :] * 50.0 + 5
        d[..., 1] = 5.0
        return d

    cases = {}
    # c0: ordinary multi-class, chain A>B>C (B suppressed by A, C overlaps B only -> survives)
    d = mk(2, 24)
    def put(d, b, r, cx, cy, w, h, conf, cls, p, extra=None):
        d[b, r, :5] = (cx, cy, w, h, conf)
        d[b, r, 5 + cls] = p
        if extra:
            for c, v in extra.items():
                d[b, r, 5 + c] = v
    put(d, 0, 0, 100, 100, 40, 40, 0.95, 3, 0.9)      # A
    put(d, 0, 1, 110, 100, 40, 40, 0.90, 3, 0.9)      # B  IOU(A,B)=0.6 -> suppressed
    put(d, 0, 2, 125, 100, 40, 40, 0.85, 3, 0.9)      # C  IOU(A,C)=0.23, IOU(B,C)=0.45 -> survives
    put(d, 0, 3, 100, 100, 40, 40, 0.99, 7, 0.9)      # same place, other class -> kept
    put(d, 0, 4, 300, 300, 20, 60, 0.8, 0, 0.7)       # class 0 lone box
    put(d, 0, 5, 300, 300, 20, 60, 0.7, 0, 0.6)       # below 0.5 after product (0.42) -> filtered
    put(d, 0, 6, 200, 50, 30, 30, 0.9, 12, 0.8, {30: 0.8})   # argmax tie -> lowest class id (12)
    put(d, 1, 7, 50, 60, 30, 20, 0.9, 79, 0.95)
    put(d, 1, 8, 52, 60, 30, 20, 0.95, 79, 0.95)      # higher score later row -> sorts first, suppresses row 7
    put(d, 1, 9, 250, 60, 0, 20, 0.99, 5, 0.99)       # zero-area -> dropped
    put(d, 1, 10, 251, 60, 30, 20, 0.9, 5, 0.9)       # not suppressed by the zero-area box
    cases["c0"] = (d, 0.5, 0.4, False, True)
    cases["c0_raw"] = (d, 0.5, 0.4, False, False)
    cases["c0_eval"] = (d, 0.3, 0.45, True, True)
    cases["c0_eval_raw"] = (d, 0.3, 0.45, True, False)
    # c1: nothing anywhere -> []
    cases["c1_empty"] = (mk(2, 8), 0.5, 0.4, False, True)
    cases["c1_empty_eval"] = (mk(2, 8), 0.5, 0.4, True, True)
    # c2: first image empty, second not -> [Tensor(0,), Tensor[n,7]]
    d = mk(3, 8)
    put(d, 1, 2, 80, 80, 30, 30, 0.9, 1, 0.9)
    cases["c2_mixed"] = (d, 0.5, 0.4, False, True)
    # c3: seeded clustered boxes, many suppressions, 3 classes, 2 images
    N = 400
    d = mk(2, N)
    u = synth.uniform01(301, 3, 2 * N * 8).reshape(2, N, 8)
    centres = np.array([[80, 80], [200, 120], [320, 300], [120, 330]], dtype=np.float32)
    k = (u[..., 0] * 4).astype(np.int64)
    d[..., 0] = centres[k][..., 0] + (u[..., 1] - 0.5) * 60
    d[..., 1] = centres[k][..., 1] + (u[..., 2] - 0.5) * 60
    d[..., 2] = 30 + u[..., 3] * 50
    d[..., 3] = 30 + u[..., 4] * 50
    d[..., 4] = 0.55 + 0.44 * u[..., 5]
    cls = (u[..., 6] * 3).astype(np.int64) * 20
    for b in range(2):
        d[b, np.arange(N), 5 + cls[b]] = (0.6 + 0.39 * u[b, :, 7])
    d[0, ::5, 4] = 0.1                                  # some rows fail the confidence filter
    cases["c3_cluster"] = (d, 0.5, 0.4, False, True)
    cases["c3_cluster_nms45_eval"] = (d, 0.4, 0.45, True, True)
    # c4: nms_thr >= 1: the diagonal is never > thr, so the reference emits nothing per class
    cases["c4_thr1"] = (cases["c0"][0], 0.5, 1.0, False, True)

    g3 = {}
    names = []
    for name, (d, ct, nt, ev, nms) in cases.items():
        res = utils.postprocessing(torch.from_numpy(d.copy()), 80, ct, nt, ev, nms)
        names.append(name)
        g3[name + "_in"] = d
        g3[name + "_cfg"] = np.array([ct, nt, float(ev), float(nms)], dtype=np.float64)
        g3[name + "_islist"] = np.array([len(res)], dtype=np.int32)      # 0 -> the [] sentinel
        for i, r in enumerate(res):
            g3["%s_out%d" % (name, i)] = r.numpy().astype(np.float32)
        print("G3", name, [tuple(r.shape) for r in res])
    g3["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "postproc.npz"), **g3)

    # ---------------------------------------------------------------- G5: end to end with SW-1
    g5 = {}
    dog = letterbox_dog(416)
    g5["dog_u8"] = dog
    dog_f = torch.from_numpy(dog.astype(np.float32) / np.float32(255.0)).permute(2, 0, 1).unsqueeze(0).contiguous()
    taps = {}

    def run(net, x):
        with torch.no_grad():
            d1, d2, d3 = net(x, None)
        return torch.cat((d1, d2, d3), 1)

    e2e = [("dog416", dog_f, None), ("u416", None, (2, 416)), ("u608", None, (1, 608))]
    for name, x, spec in e2e:
        seed = 11 if name == "u416" else 12
        best = None
        for trial in range(1 if spec is None else 12):
            if spec is not None:
                x = torch.from_numpy(synth.images(spec[0], spec[1], seed + 100 * trial))
            net.img_dim = (x.shape[3], x.shape[2])
            dets = run(net, x)
            mg = margins(torch, utils, dets, 0.5, 0.4)
            # robustness score: every decision margin relative to the fp32 noise it must survive
            score = min(mg["conf"] / 1e-4, mg["cls"] / 1e-5, mg["iou"] / 5e-4, mg["tie"] / 1e-6)
            print("G5", name, "seed", seed + 100 * trial, mg, "score %.2f" % score)
            if best is None or score > best[0]:
                best = (score, seed + 100 * trial, dets, mg, x)
        score, seed, dets, mg, x = best
        print("G5", name, "picked seed", seed, "score %.2f" % score)
        # per-scale logits statistics come from hooks on the three plain head convs
        res = utils.postprocessing(dets.clone(), 80, 0.5, 0.4)
        g5[name + "_seed"] = np.array([seed], dtype=np.int64)
        g5[name + "_margins"] = np.array([mg["conf"], mg["cls"], mg["iou"], mg["tie"]], dtype=np.float64)
        rows = np.arange(0, dets.shape[1], 37, dtype=np.int32)
        g5[name + "_rows"] = rows
        g5[name + "_dets_rows"] = dets[:, rows].numpy()
        g5[name + "_dets_sum"] = np.array([float(dets.double().sum()), float(dets.double().abs().max())])
        g5[name + "_nres"] = np.array([len(res)], dtype=np.int32)
        for i, r in enumerate(res):
            g5["%s_boxes%d" % (name, i)] = r.numpy().astype(np.float32)
            print("   image", i, "boxes", tuple(r.shape))
        # the candidate rows (pre-NMS) so tests can check the filter separately
        d = dets.clone()
        sc = d[..., 5:] * d[..., 4:5]
        mx, arg = sc.max(-1)
        cand = (mx > 0.5).nonzero()
        g5[name + "_cand"] = torch.cat((cand, arg[mx > 0.5].unsqueeze(1)), 1).numpy().astype(np.int32)
    # eval mode (is_eval=True, nms 0.45 as evaluate.py:203) on the dog image.  The reference's 0.005
    # confidence floor passes ~260k (row, class) pairs with synthetic weights, so the fixture uses 0.4.
    net.img_dim = (416, 416)
    dets = run(net, dog_f)
    res = utils.postprocessing(dets.clone(), 80, 0.4, 0.45, True, True)
    g5["dog416_eval_boxes0"] = res[0].numpy().astype(np.float32)
    g5["dog416_eval_cfg"] = np.array([0.4, 0.45])
    print("G5 dog eval boxes", tuple(res[0].shape))
    np.savez_compressed(os.path.join(GOLD, "e2e.npz"), **g5)

    # ---------------------------------------------------------------- G7: neighbours of the path
    g7 = {}
    raw = synth.uniform(701, 1, 60 * 4, -40.0, 460.0).reshape(60, 4)
    boxes = np.stack((np.minimum(raw[:, 0], raw[:, 2]), np.minimum(raw[:, 1], raw[:, 3]),
                      np.maximum(raw[:, 0], raw[:, 2]), np.maximum(raw[:, 1], raw[:, 3])), 1).astype(np.float32)
    boxes[7] = 0.0                                    # all-zero row: left untouched by the reference (mask)
    g7["boxes"] = boxes
    cases = [(602, 452, 416, 416), (640, 480, 608, 608), (333, 500, 416, 416), (1920, 1080, 416, 416), (416, 416, 416, 416)]
    g7["cases"] = np.array(cases, dtype=np.int32)
    for ci, (ow, oh, iw, ih) in enumerate(cases):
        for lb in (0, 1):
            out = boundingbox.correct_yolo_boxes(torch.from_numpy(boxes.copy()), ow, oh, iw, ih, bool(lb))
            g7["out_%d_%d" % (ci, lb)] = out.numpy().astype(np.float32)
        g7["xyxy_%d" % ci] = boundingbox.letterbox_reverse(torch.from_numpy(boxes.copy()), ow, oh, iw, ih).numpy()
        g7["rescale_%d" % ci] = boundingbox.rescale_bbox(torch.from_numpy(boxes.copy()), ow, oh, iw, ih).numpy()
        g7["trans_%d" % ci] = np.array(utils.letterbox_transforms((ow, oh), (iw, ih)), dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, "neighbours.npz"), **g7)
    print("G7 done")

    # ---------------------------------------------------------------- G6: per-layer probes (bring-up aid)
    stats = {}
    hooks = []
    order = []

    def hook(name):
        def f(m, i, o):
            o = o.detach()
            stats[name] = [float(o.mean()), float(o.std()), float(o.abs().max())]
            order.append(name)
        return f

    for n, m in net.named_modules():
        if type(m).__name__ == "conv_bn_relu" or (isinstance(m, torch.nn.Conv2d) and n.endswith("mlist.6")):
            hooks.append(m.register_forward_hook(hook(n)))
    run(net, torch.from_numpy(synth.images(1, 416, 11)))
    for h in hooks:
        h.remove()
    meta["layer_stats_u416_seed11_b1"] = stats
    meta["layer_order"] = order
    json.dump(meta, open(os.path.join(GOLD, "meta.json"), "w"), indent=0)
    print("done")


if __name__ == "__main__":
    main()
