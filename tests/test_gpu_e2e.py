"""End-to-end GPU parity: YoloNet.forward / detect() against the reference's golden outputs
(tests/golden/e2e.npz, produced by the reference itself with SW-1 weights) and the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle_cpu as oc
from yolo_v3_amd import synth, detect, postprocessing, Detector, _ffi
from tests.helpers import TOL, assert_close_rel, match_boxes, check_result_convention, load_sw1_net, detector_dets, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net(sw1_stream):
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return load_sw1_net(sw1_stream).cuda()


def _input(g, name):
    if name == "dog416":
        return torch.from_numpy(g["dog_u8"].astype(np.float32) / np.float32(255.0)).permute(2, 0, 1).unsqueeze(0).contiguous()
    b, s = {"u416": (2, 416), "u608": (1, 608)}[name]
    return torch.from_numpy(synth.images(b, s, int(g[name + "_seed"][0])))


@pytest.mark.parametrize("mode", [_ffi.F32, _ffi.F32X3, _ffi.F32H2])
@pytest.mark.parametrize("name", ["dog416", "u416", "u608"])
def test_forward_and_boxes_vs_reference_golden(golden_dir, net, name, mode):
    """BASELINE configs[0]/[1]-shaped cases, in both fp32 math modes (exact fp32 MFMA and the bf16x3
    split).  Tolerance: 1e-4 * max(1,|ref|) on every detection value and every final box column
    (north-star: "within 1e-4 fp32"); counts and classes exact."""
    net.math_mode = mode
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    x = _input(g, name).cuda()
    with torch.no_grad():
        d1, d2, d3 = net(x, None)
    hw = x.shape[2] // 32
    assert d1.shape == (x.shape[0], hw * hw * 3, 85) and d3.shape[1] == 16 * d1.shape[1]
    dets = torch.cat((d1, d2, d3), 1)
    rows = g[name + "_rows"]
    err = assert_close_rel(dets[:, rows].cpu(), g[name + "_dets_rows"], TOL, name + " detections")
    # candidate set (pre-NMS filter decisions) identical to the reference's
    sc = dets[..., 5:] * dets[..., 4:5]
    mx, arg = sc.max(-1)
    cand = torch.cat(((mx > 0.5).nonzero(), arg[mx > 0.5].unsqueeze(1)), 1).cpu().numpy().astype(np.int32)
    assert np.array_equal(cand, g[name + "_cand"]), "candidate (image,row,class) set differs"
    # the reference idiom and the fused detect() agree with each other and with the golden boxes
    res = postprocessing(dets, 80, 0.5, 0.4)
    fused = detect(net, x, 80, 0.5, 0.4)
    assert len(res) == len(fused) == int(g[name + "_nres"][0])
    worst = 0.0
    one_plan = list(net._detectors.values())[-1].lanes == 1      # the detector ran the same one-lane plan as net(x): same bits
    for i, (r, f) in enumerate(zip(res, fused)):
        if one_plan:
            assert torch.equal(r, f)
        else:                                                     # (forced YV3_LANES=2 on a tiny batch: another schedule per lane)
            match_boxes(f, r, TOL)
        worst = max(worst, match_boxes(r, g["%s_boxes%d" % (name, i)], TOL), match_boxes(f, g["%s_boxes%d" % (name, i)], TOL))
    print("%s mode %d: max detection err %.3g, max box err %.3g" % (name, mode, err, worst))


def test_eval_mode_vs_reference_golden(golden_dir, net):
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    x = _input(g, "dog416").cuda()
    ct, nt = g["dog416_eval_cfg"]
    res = detect(net, x, 80, float(ct), float(nt), is_eval=True)
    exp = g["dog416_eval_boxes0"]
    # ~1900 boxes: a handful sit within fp32 noise of a threshold, so compare as sets with a small budget
    got = res[0].double()
    exp_t = torch.from_numpy(exp).double()
    assert abs(len(got) - len(exp_t)) <= 4
    scale = torch.ones_like(exp_t[:, :6])
    scale[:, :4] = exp_t[:, :4].abs().amax(1, keepdim=True).clamp(min=1.0)      # see helpers.match_boxes
    d = ((exp_t[:, None, :6] - got[None, :, :6]).abs() / scale[:, None]).amax(-1)
    d[exp_t[:, 6][:, None] != got[None, :, 6]] = 1e9
    unmatched = int((d.min(1)[0] > TOL).sum())
    assert unmatched <= 4, "%d of %d reference boxes have no match within 1e-4" % (unmatched, len(exp_t))


def test_decisions_exact_on_identical_detections(net):
    """Filter / sort / NMS are integer-and-compare work: on the GPU's own detections the HIP
    post-processing must equal the oracle's bit for bit (B=4, 416)."""
    x = torch.from_numpy(synth.images(4, 416, 4242)).cuda()
    res = detect(net, x)
    ref = oc.postprocess(detector_dets(net).cpu(), 80, 0.5, 0.4)
    check_result_convention(res, ref)
    for r, e in zip(res, ref):
        assert torch.equal(r, e)
    res = detect(net, x, 80, 0.3, 0.45, is_eval=True)
    ref = oc.postprocess(detector_dets(net).cpu(), 80, 0.3, 0.45, True, True)
    for r, e in zip(res, ref):
        assert torch.equal(r, e)


@pytest.mark.parametrize("mode", [_ffi.F32, _ffi.F32X3, _ffi.F32H2])
def test_conv_trunk_vs_oracle_layerwise(net, sw1_stream, mode):
    """Per-layer bring-up check: every conv output of the plan vs the oracle's tap (B=1, 416), in the
    exact-fp32 MFMA mode and in the bf16x3-split mode (same tolerance: both are fp32-class arithmetic)."""
    from yolo_v3_amd import engine
    x = torch.from_numpy(synth.images(1, 416, 11))
    sd, _ = oc.state_dict_from_stream(sw1_stream)
    taps = []
    with torch.no_grad():
        oc.head_logits(sd, x, taps)
        eng = net.engine(mode)
        eng.fuse_decode, eng.fuse_front, eng._plans = False, False, {}     # materialise the head logits and the first layers' outputs
        try:
            _, plan = eng.forward(x.cuda())
            torch.cuda.synchronize()
        finally:
            eng.fuse_decode, eng.fuse_front, eng._plans = True, True, {}
    assert len(taps) == 75
    worst = 0.0
    for name, ref in taps:
        got = engine.from_planes(plan.layer_out[name], mode).permute(0, 3, 1, 2).float().cpu()
        worst = max(worst, assert_close_rel(got, ref, TOL, name))
    print("mode %d: worst layer error %.3g" % (mode, worst))


@pytest.mark.parametrize("name", ["dog416", "u416", "u608"])
def test_split_mode_boxes_vs_reference_golden(golden_dir, net, name):
    """The throughput mode (YV3_F32_BF16X3) against the reference's golden detections and boxes, at the
    same 1e-4 tolerance as the exact-fp32 mode."""
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    x = _input(g, name).cuda()
    with torch.no_grad():
        dets = net.forward_cat(x, dtype=_ffi.F32X3)
    rows = g[name + "_rows"]
    err = assert_close_rel(dets[:, rows].cpu(), g[name + "_dets_rows"], TOL, name + " detections (split mode)")
    res = postprocessing(dets, 80, 0.5, 0.4)
    assert len(res) == int(g[name + "_nres"][0])
    worst = 0.0
    for i, r in enumerate(res):
        worst = max(worst, match_boxes(r, g["%s_boxes%d" % (name, i)], TOL))
    det = Detector(net, x.shape[0], x.shape[2], x.shape[3], dtype=_ffi.F32X3)
    for r, f in zip(res, det(x)):
        assert torch.equal(r, f)
    print("%s split mode: max detection err %.3g, max box err %.3g" % (name, err, worst))


def test_bf16_mode_config3_shape(net):
    """BASELINE configs[2] shape (608x608, bf16 convs / fp32 decode), B=4.  bf16 is the reduced-precision
    throughput mode, outside the 1e-4 bar: detections must stay close to the exact-fp32 mode on average
    (mean normalised error < 5e-3) and most boxes must survive with matching class and IOU > 0.9."""
    x = torch.from_numpy(synth.images(4, 608, 77)).cuda()
    with torch.no_grad():
        d32 = net.forward_cat(x, dtype=_ffi.F32)
        dbf = net.forward_cat(x, dtype=_ffi.BF16)
    assert dbf.shape == (4, 22743, 85)
    e = (dbf - d32).abs() / d32.abs().clamp(min=1.0)
    assert float(e.mean()) < 5e-3, float(e.mean())
    r32, rbf = postprocessing(d32, 80, 0.5, 0.4), postprocessing(dbf, 80, 0.5, 0.4)
    n32, matched = 0, 0
    for a, b in zip(r32, rbf):
        n32 += len(a)
        if len(a) and len(b):
            iou = bbox_iou_gpu(a[:, :4], b[:, :4])
            same = a[:, 6][:, None] == b[:, 6][None, :]
            matched += int(((iou > 0.9) & same).any(1).sum())
    assert n32 > 20 and matched >= 0.6 * n32, (n32, matched)     # bf16 noise moves boxes whose w,h = exp(t)*anchor are huge


def bbox_iou_gpu(a, b):
    from yolo_v3_amd import bbox_iou
    return bbox_iou(a.cuda(), b.cuda()).cpu()


def test_full_size_properties(net, sw1_stream):
    """BASELINE configs[1] size (32 x 416 x 416): independence of the position in the batch (duplicated / permuted images give
    identical boxes bit for bit), agreement of eager vs HIP-graph replay, and independence of the batch SIZE: bit for bit
    with the direct kernels only (``net.winograd = False``); with the default per-launch choice between the direct and the
    Winograd form of a layer (it depends on the tile count, i.e. on the batch size) two batch sizes may run different
    forms of the same layer -- same boxes within the parity tolerance."""
    base = synth.images(8, 416, 99)
    idx = [0, 1, 2, 3, 4, 5, 6, 7] * 4
    x = torch.from_numpy(base[idx]).cuda()
    det = Detector(net, 32, 416, 416)
    res = det(x)
    assert len(res) == 32
    for i in range(8, 32):
        assert tuple(res[i].shape) == tuple(res[i % 8].shape) and torch.equal(res[i], res[i % 8])
    ref = detect(net, torch.from_numpy(base[:2]).cuda())
    for i in range(2):
        match_boxes(res[i], ref[i])                           # another batch size: same boxes within 1e-4
    gdet = Detector(net, 32, 416, 416, graph=True)
    for _ in range(2):
        res_g = gdet(x)
    for a, b in zip(res, res_g):
        assert torch.equal(a, b)
    direct = load_sw1_net(sw1_stream).cuda()
    direct.winograd, direct.stream_k = False, False             # (stream_k None would switch the stream-K schedule on for single images)
    res_d = Detector(direct, 32, 416, 416)(x)
    ref_d = detect(direct, torch.from_numpy(base[:2]).cuda())
    for i in range(2):
        assert torch.equal(res_d[i], ref_d[i])                # direct kernels: independent of batch size / position, bit for bit
    for i in range(8, 32):
        assert torch.equal(res_d[i], res_d[i % 8])


@pytest.mark.parametrize("B,lanes", [(128, 1), (96, 2)])
def test_large_batches_take_the_winograd_form_and_keep_the_boxes(net, sw1_stream, B, lanes):
    """Above 64 images per GPU the per-launch rule selects the Winograd form by the fill of the LAST round of tiles (one lane,
    bs=128: 2.64 and 1.53 rounds) or because the chip is shared (two lanes, 48 images each: YV3_OPT_TWO_LANES): the boxes equal
    those of the direct kernels (``net.winograd = False``) within the 1e-4 parity tolerance, image for image, and the
    detections tensors agree within the same tolerance."""
    base = synth.images(16, 416, 123)
    x = torch.from_numpy(base[[i % 16 for i in range(B)]]).cuda()
    det = Detector(net, B, 416, 416, lanes=lanes)
    res = det(x)
    dets_w = det.dets.clone()
    direct = load_sw1_net(sw1_stream).cuda()
    direct.winograd, direct.stream_k = False, False
    ddet = Detector(direct, B, 416, 416, lanes=lanes)
    res_d = ddet(x)
    assert len(res) == len(res_d) == B
    assert not torch.equal(dets_w, ddet.dets), "the Winograd form was expected to run for some layers at this batch size"
    assert_close_rel(dets_w, ddet.dets, TOL, "Winograd vs direct detections")
    for a, b in zip(res[:32], res_d[:32]):
        match_boxes(a, b)


def test_weights_change_is_picked_up(net, sw1_stream):
    """Packed weights follow the parameters (load_state_dict / in-place edits), like eager modules."""
    x = torch.from_numpy(synth.images(1, 416, 5)).cuda()
    a = net.forward_cat(x).clone()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        net.pre_det1.mlist[6].bias.add_(1.0)
    b = net.forward_cat(x).clone()
    assert not torch.equal(a[:, :507], b[:, :507]) and torch.equal(a[:, 507:], b[:, 507:])
    net.load_state_dict(sd)
    assert torch.equal(net.forward_cat(x), a)


def test_predict_idiom_matches_oracle_composition(net):
    """reference test.py:28-46 end to end for images of different sizes: letterbox -> net -> NMS -> boxes in
    the ORIGINAL image.  Checked against the oracle composition on the SAME letterboxed input: classes/counts
    exact, xywh within 1e-4 * image size."""
    from yolo_v3_amd import predict, letterbox_batch
    net.math_mode = _ffi.F32X3
    base = (synth.images(2, 416, 123) * 255).astype(np.uint8)
    imgs = [np.ascontiguousarray(base[0].transpose(1, 2, 0)[:300, :]), np.ascontiguousarray(base[1].transpose(1, 2, 0)[:, :250])]
    preds = predict(net, imgs, (416, 416), obj_conf_thr=0.1)
    batch, _ = letterbox_batch(imgs, (416, 416))
    sd_boxes = detect(net, batch, obj_conf_thr=0.1)
    assert len(preds) == 2 and len(sd_boxes) == 2 and min(len(b) for b in sd_boxes) > 0
    for i, im in enumerate(imgs):
        ref = oc.correct_yolo_boxes(sd_boxes[i][:, :4], im.shape[1], im.shape[0], 416, 416, True)
        assert preds[i].shape == (len(sd_boxes[i]), 5)
        assert torch.equal(preds[i][:, 0], sd_boxes[i][:, 6])
        assert torch.equal(preds[i][:, 1:], ref)                    # same fp32 ops on the same boxes -> bitwise
        assert float(preds[i][:, 1].min()) >= 0 and float((preds[i][:, 1] + preds[i][:, 3]).max()) <= im.shape[1] + 1e-3
    # is_letterbox=False (test.py:28,41): plain cv2.resize to the network size, boxes back through rescale_bbox
    from yolo_v3_amd import resize_batch
    preds = predict(net, imgs, (416, 416), obj_conf_thr=0.1, is_letterbox=False)
    rbatch = resize_batch(imgs, (416, 416))
    for i, im in enumerate(imgs):
        assert torch.equal(rbatch[i].cpu(), oc.resize_image(im, (416, 416)))
    rboxes = detect(net, rbatch, obj_conf_thr=0.1)
    for i, im in enumerate(imgs):
        ref = oc.correct_yolo_boxes(rboxes[i][:, :4], im.shape[1], im.shape[0], 416, 416, False)
        assert preds[i].shape == (len(rboxes[i]), 5) and torch.equal(preds[i][:, 0], rboxes[i][:, 6])
        assert torch.equal(preds[i][:, 1:], ref)


@pytest.mark.parametrize("nc,size,mode", [(20, 320, _ffi.F32X3), (1, 352, _ffi.F32), (20, 320, _ffi.F32), (20, 320, _ffi.F32H2), (1, 352, _ffi.F32H2)])
def test_other_class_counts_and_sizes(nc, size, mode):
    """Custom-data shapes (reference README: VOC 20 classes, x-wing 1 class): head width 3*(5+nc) is no longer
    255, input size is not 416/608.  Whole net vs the oracle at 1e-4, decisions exact on identical detections."""
    from yolo_v3_amd import YoloNet, WeightManager
    stream = synth.weight_stream(num_class=nc, seed=77)
    net = YoloNet((size, size), numClass=nc).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    net = net.cuda()
    net.math_mode = mode
    x = torch.from_numpy(synth.images(2, size, 9))
    sd, _ = oc.state_dict_from_stream(stream, nc)
    with torch.no_grad():
        ref = torch.cat(oc.yolonet_forward(sd, x, num_class=nc), 1)
        dets = net.forward_cat(x.cuda())
    assert dets.shape == ref.shape == (2, 3 * 21 * (size // 32) ** 2, 5 + nc)
    assert_close_rel(dets.cpu(), ref, TOL, "nc=%d size=%d" % (nc, size))
    thr = 0.3
    got = detect(net, x.cuda(), nc, thr, 0.4)
    exp = oc.postprocess(detector_dets(net).cpu(), nc, thr, 0.4)
    check_result_convention(got, exp)
    for a, b in zip(got, exp):
        assert torch.equal(a, b)


def test_fp16_plane_overflow_falls_back_to_bf16x3(sw1_stream):
    """the default mode computes where the reference computes.  A residual branch whose 1x1 output is scaled by 2^17
    (BN scale and shift x 2^17, the following 3x3 weights x 2^-17: the same function, exactly, in fp32) carries activations of ~1e6:
    beyond the fp16 planes of F32H2.  `detect` / `net(x)` notice the kernels' saturation flag, re-run the batch in F32X3 (three bf16
    planes, fp32 range) with ONE RuntimeWarning, return the oracle's boxes within 1e-4, and the network stays in the fall-back mode
    (no second warning, no second discarded pass)."""
    import warnings
    from oracle import oracle_cpu as oc
    from oracle.boxdelta import boxes_delta
    net = load_sw1_net(sw1_stream).cuda()
    x = torch.from_numpy(synth.images(2, 416, 3))
    with torch.no_grad():
        blk = net.feature.mlist[4]
        blk.conv1.bn.weight.mul_(2.0 ** 17); blk.conv1.bn.bias.mul_(2.0 ** 17)
        blk.conv2.conv.weight.mul_(2.0 ** -17)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        taps = []
        oc.head_logits(sd, x, taps)
        assert max(float(t.abs().max()) for n, t in taps if n == "feature.mlist.4.conv1") > 65504.0
        ref = torch.cat(oc.yolonet_forward(sd, x), 1)
    want = oc.postprocess(ref.clone(), 80, 0.5, 0.4)
    assert net.math_mode == _ffi.F32H2
    with pytest.warns(RuntimeWarning, match="fp16 range") as rec:
        res = detect(net, x.cuda())
    assert len([w for w in rec if "fp16 range" in str(w.message)]) == 1
    d = boxes_delta(res, want, 2)
    print("saturation fall-back: boxes", d)
    assert d["ref_boxes"] > 10 and d["unmatched_frac"] <= 0.005
    assert d["max_rel_err_coords"] <= TOL and d["max_abs_err_conf"] <= TOL and d["max_abs_err_score"] <= TOL
    assert net.engine().dtype == _ffi.F32X3                       # remembered: F32H2 requests now get the fall-back engine
    with warnings.catch_warnings():
        warnings.simplefilter("error")                            # no second warning
        res2 = detect(net, x.cuda())
        dets = net.forward_cat(x.cuda()).cpu()
    assert all(torch.equal(a, b) for a, b in zip(res, res2))
    assert_close_rel(dets, ref, TOL, "fall-back detections")
    # the forward path notices it by itself too (fresh network, no detect() before)
    net2 = load_sw1_net(sw1_stream).cuda()
    net2.load_state_dict(net.state_dict())
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        d1, d2, d3 = net2(x.cuda())
    assert_close_rel(torch.cat((d1, d2, d3), 1).cpu(), ref, TOL, "fall-back forward")


def test_fp16_plane_overflow_is_reported(sw1_stream):
    """The strict form (``net.strict_range = True``, the behaviour before round 6): F32H2 stores activations as fp16 hi+lo planes,
    a network whose activations leave +-65504 must not silently return saturated results.  Blow up one BN scale -> detect()
    raises; the other modes still work."""
    net = load_sw1_net(sw1_stream).cuda()
    net.strict_range = True
    x = torch.from_numpy(synth.images(1, 416, 3)).cuda()
    net.math_mode = _ffi.F32H2
    detect(net, x)                                               # sane weights: fine
    with torch.no_grad():
        net.feature.mlist[3].bn.weight.mul_(1e6)
    with pytest.raises(_ffi.Yv3Error, match="fp16 range"):
        detect(net, x)
    with pytest.raises(_ffi.Yv3Error, match="fp16 range"):      # asynchronous path: reported by a later forward
        for _ in range(3):
            net.forward_cat(x)
            torch.cuda.synchronize()
    net.math_mode = _ffi.F32X3                                   # bf16 planes have fp32's exponent range: no error
    out = net.forward_cat(x)
    assert out.shape == (1, 10647, 85) and torch.isfinite(out[..., 4:]).all()


# ----------------------------------------------------------------------------- COCO results writer
def _coco_inputs(golden_dir):
    g = np.load(os.path.join(golden_dir, "coco_results_inputs.npz"))
    org, counts, flat, paths = g["org"], g["counts"], g["preds"], [str(p) for p in g["paths"]]
    preds, k = [], 0
    for n in counts:
        preds.append(torch.from_numpy(flat[k:k + n].copy()) if n else torch.Tensor())
        k += n
    sample = {"img": torch.zeros(len(org), 3, 416, 416), "org_img": [torch.zeros(3, int(h), int(w)) for (w, h) in org],
              "img_path": paths}
    return sample, preds


@pytest.mark.gpu
@pytest.mark.parametrize("lb", [0, 1])
def test_coco_results_writer_matches_reference_bytes(golden_dir, tmp_path, lb):
    """yolo_v3_amd.evaluate.JsonPredictionWriter (boxes through yv3_correct_boxes) writes the byte-identical file
    the reference's evaluate.JsonPredictionWriter wrote for the same predictions (oracle/make_golden_coco.py)."""
    from yolo_v3_amd import evaluate
    sample, preds = _coco_inputs(golden_dir)
    out = str(tmp_path / "res.json")
    with evaluate.open_json_pred_writer(out, None, bool(lb)) as wr:
        wr.process_batch(sample, preds)
    want = open(os.path.join(golden_dir, "coco_results_lb%d.json" % lb)).read()
    assert open(out).read() == want
    assert len(json.loads(want)) == 13


@pytest.mark.gpu
def test_predict_and_process_eval_mode(tmp_path, net):
    """reference evaluate.py:197-206 end to end on synthetic scenes: eval-mode detections handed to a BatchHandler
    equal detect()'s at the same thresholds (conf 0.4 here: with synthetic weights the reference's 0.005 floor
    passes > 5e5 (row, class) pairs per image), and the JSON written from them parses, one entry per box."""
    from yolo_v3_amd import evaluate
    net.img_dim = (416, 416)
    x = torch.from_numpy(synth.images(2, 416, 77))
    sample = {"img": x, "org_img": [torch.zeros(3, 480, 640), torch.zeros(3, 333, 500)],
              "img_path": ["a/COCO_val2014_000000000042.jpg", "b/000000000007.jpg"]}

    class Rec(evaluate.BatchHandler):
        def process_batch(self, sample, predictions):
            self.n = [int(p.shape[0]) if p.numel() else 0 for p in predictions]
            self.pred = predictions
    rec = Rec()
    evaluate.predict_and_process([sample], net, 80, rec, obj_conf_thr=0.4)
    want = detect(net, x.cuda(), 80, 0.4, 0.45, True, True)
    assert rec.n == [int(p.shape[0]) for p in want] and sum(rec.n) > 0
    out = str(tmp_path / "r.json")
    with evaluate.open_json_pred_writer(out, None, True) as wr:
        wr.process_batch(sample, [p[:50] for p in rec.pred])
    js = json.load(open(out))
    assert len(js) == sum(min(n, 50) for n in rec.n)
    assert {e["image_id"] for e in js} <= {42, 7} and all(len(e["bbox"]) == 4 for e in js)
    for e in js:
        w, h = (640, 480) if e["image_id"] == 42 else (500, 333)
        assert 0 <= e["bbox"][0] <= w and 0 <= e["bbox"][1] <= h and e["bbox"][0] + e["bbox"][2] <= w + 1e-3


@pytest.mark.gpu
def test_stream_k_schedule_end_to_end(golden_dir, sw1_stream):
    """net.stream_k = True (opt-in persistent schedule of the 13x13 layers): detections at bs=64 stay within the
    parity tolerance of the default schedule's (which is golden-checked above) -- bitwise equality is NOT expected,
    a split tile is summed head + tail -- and the kept boxes agree."""
    x = torch.from_numpy(synth.images(64, 416, 4242)).cuda()
    a = load_sw1_net(sw1_stream).cuda()
    b = load_sw1_net(sw1_stream).cuda()
    b.stream_k = True
    da, db = a.forward_cat(x).cpu(), b.forward_cat(x).cpu()
    assert b.engine().plan(64, 416, 416).workspace is not None and a.engine().plan(64, 416, 416).workspace is None
    assert_close_rel(db, da.double(), TOL, "stream-K vs default schedule")      # two fp32-round-off paths, 75 layers deep
    ra, rb = detect(a, x), detect(b, x)
    assert [int(r.shape[0]) for r in ra] == [int(r.shape[0]) for r in rb]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [_ffi.F32X3, _ffi.F32H2, _ffi.BF16])
def test_fused_decode_equals_separate_decode_bitwise(net, mode):
    """The YOLO decode fused into the head convs' epilogue (default for the plane modes: the logits are never
    written) produces bit-identical detections to head conv -> yv3_decode, at a 416 and a non-square size."""
    for (B, H, W, seed) in ((3, 416, 416, 31), (2, 320, 480, 32)):
        x = torch.from_numpy(synth.images(B, max(H, W), seed)[:, :, :H, :W].copy()).cuda()
        eng = net.engine(mode)
        net.img_dim = (W, H)
        try:
            eng.fuse_decode, eng._plans = True, {}
            a, pa = eng.forward(x)
            a = a.clone()
            assert pa.fused_decode and all(lg is None for (lg, _, _) in pa.logits)
            eng.fuse_decode, eng._plans = False, {}
            b, pb = eng.forward(x)
            assert not pb.fused_decode
            assert torch.equal(a, b)
        finally:
            eng.fuse_decode, eng._plans = True, {}
            net.img_dim = (416, 416)


@pytest.mark.gpu
@pytest.mark.parametrize("stream_k", [False, True])
def test_odd_shapes_vs_oracle(sw1_stream, stream_k):
    """Whole-net detections vs the CPU oracle on batch sizes / image sizes that exercise every tile configuration's
    tails (M not a multiple of any tile, non-square grids, a grid smaller than one tile), for both schedules."""
    sd, _ = oc.state_dict_from_stream(sw1_stream)
    net = load_sw1_net(sw1_stream).cuda()
    net.stream_k = stream_k
    for (B, H, W) in [(3, 320, 320), (5, 352, 608), (2, 96, 64), (9, 224, 416)]:
        net.img_dim = (W, H)
        x = torch.from_numpy(synth.images(B, max(H, W), 100 + B)[:, :, :H, :W].copy())
        with torch.no_grad():
            got = net.forward_cat(x.cuda()).cpu()
            ref = torch.cat(oc.yolonet_forward(sd, x), 1)
        assert_close_rel(got, ref.double(), TOL, "B=%d %dx%d stream_k=%d" % (B, H, W, stream_k))



@pytest.mark.gpu
def test_batch_split_schedule_is_bit_identical(sw1_stream):
    """Opt-in schedule (net.batch_split): at bs=64 the 13x13 3x3 layers (344 tiles of 256x128 = 1.34 rounds of the chip) run
    as two launches over batch slices, 48 + 16 images (yv3_conv_desc.*_plane_stride): same kernels, same K order ->
    detections bit-identical to the one-launch plan; the plan really contains the extra launches."""
    net = load_sw1_net(sw1_stream).cuda()
    net.winograd = False                                     # (a batch-split plan keeps the direct kernels: compare like with like)
    x = torch.from_numpy(synth.images(64, 416, 4243)).cuda()
    eng = net.engine()
    outs = []
    for on in (False, True):
        eng.batch_split, eng._plans = on, {}
        try:
            d, plan = eng.forward(x)
            outs.append((d.clone(), plan.n_desc))
        finally:
            eng.batch_split, eng._plans = False, {}
    assert outs[0][1] == 74 and outs[1][1] == 74 + 8          # the 512->1024 3x3 layers: 4 res blocks + 3 branch convs + the s2 conv
    assert torch.equal(outs[0][0], outs[1][0])


@pytest.mark.gpu
@pytest.mark.parametrize("B,size", [(64, 416), (17, 320)])
def test_two_lanes_are_bit_identical_to_one(sw1_stream, B, size):
    """Detector(lanes=2): the batch as two contiguous sub-batches on two HIP streams (same kernels, same K order, one shared
    status word) -> detections and final boxes bit-identical to the single-lane run, for an even and an odd batch; the
    default (lanes=None) calibrates the stream pair and ends up with 1 or 2 lanes, same results either way; saturation in
    EITHER lane is reported."""
    net = load_sw1_net(sw1_stream, size).cuda()
    net.winograd = False            # lane mechanics: the same (direct) kernels on both sides; the per-launch Winograd choice depends
                                    # on the sub-batch size and is compared within tolerance at the end
    x = torch.from_numpy(synth.images(B, size, 77)).cuda()
    one, two, auto = (Detector(net, B, size, size, lanes=n) for n in (1, 2, None))
    assert (one.lanes, two.lanes) == (1, 2) and auto.lanes in (1, 2)
    assert [p.B for p in two.lane_plans] == [(B + 1) // 2, B // 2]
    r1, r2, ra = one(x), two(x), auto(x)
    assert torch.equal(one.dets, two.dets) and torch.equal(one.dets, auto.dets)
    assert len(r1) == len(r2) == len(ra) == B and sum(len(b) for b in r1) > B
    for a, b, c in zip(r1, r2, ra):
        assert tuple(a.shape) == tuple(b.shape) and torch.equal(a, b) and torch.equal(a, c)
    bad = x.clone()
    bad[B - 1] *= 1e4                                           # an image of the SECOND lane leaves the fp16 range of the first layer
    net.strict_range = True                                     # (the default would move the detector to F32X3: test_fp16_plane_overflow_falls_back_...)
    with pytest.raises(_ffi.Yv3Error, match="fp16 range"):
        two(bad)
    assert all(torch.equal(a, b) for a, b in zip(two(x), r1))   # and the detector is usable again afterwards
    if B == 64:
        # default network (Winograd form chosen per launch: at 64 images one lane runs the 13x13 layers in it, two lanes of 32 the
        # 26x26 layers): the two schedules agree within the parity tolerance
        wnet = load_sw1_net(sw1_stream, size).cuda()
        w1, w2 = Detector(wnet, B, size, size, lanes=1)(x), Detector(wnet, B, size, size, lanes=2)(x)
        for a, b in zip(w1, w2):
            match_boxes(a, b)


@pytest.mark.gpu
def test_repack_reaches_a_held_detector(sw1_stream):
    """A write through ``param.data`` (the reference loader's idiom, darknet.py:275) is invisible to the (data_ptr, _version)
    signature; ``net.repack()`` must invalidate the engine IN PLACE so that a Detector the caller already holds runs the
    new weights (it used to keep an orphaned engine and silently run the stale ones).  Also: the position-sensitive
    checksum mode sees a swap of two filters, which preserves every per-tensor L1 norm."""
    net = load_sw1_net(sw1_stream).cuda()
    x = torch.from_numpy(synth.images(2, 416, 6)).cuda()
    det = Detector(net, 2, 416, 416)
    det(x)
    a = det.dets.clone()
    bias = net.pre_det1.mlist[6].bias
    bias.data.copy_(bias.data + 1.0)                          # does NOT bump bias._version
    det(x)
    assert torch.equal(det.dets, a)                           # documented: invisible without repack()
    net.repack()
    det(x)
    assert not torch.equal(det.dets[:, :507], a[:, :507]) and torch.equal(det.dets[:, 507:], a[:, 507:])
    assert torch.equal(net.forward_cat(x), det.dets)          # and net.forward sees the same engine state
    # checksum mode: swap two filters of one conv through .data -- same L1 norm, different network
    net.weight_check = "checksum"
    b = net.forward_cat(x).clone()
    w = net.feature.mlist[1].conv.weight
    tmp = w.data[0].clone(); w.data[0].copy_(w.data[1]); w.data[1].copy_(tmp)
    c = net.forward_cat(x)
    assert not torch.equal(b, c)


@pytest.mark.gpu
def test_eval_detect_fused_equals_two_phase(sw1_stream):
    """detect(is_eval=True) runs the fused Detector (16 384 candidate slots per image) and falls back to the two-phase
    forward_cat -> postprocessing path on overflow; both give the same boxes, bit for bit, and alternating two batch shapes
    re-uses the cached detectors (no re-allocation / re-calibration)."""
    net = load_sw1_net(synth.eval_weight_stream()).cuda()
    x = torch.from_numpy(synth.images(3, 416, 8)).cuda()
    fused = detect(net, x, 80, 0.005, 0.45, is_eval=True)
    two_phase = postprocessing(net.forward_cat(x), 80, 0.005, 0.45, True, True)
    assert len(fused) == len(two_phase) == 3 and sum(len(b) for b in fused) > 100
    for a, b in zip(fused, two_phase):
        assert tuple(a.shape) == tuple(b.shape) and torch.equal(a, b)
    # overflow of the fused buffers -> same answer through the fallback
    import importlib
    dmod = importlib.import_module("yolo_v3_amd.detect")
    key = [k for k in net._detectors if k[4] is True][0]
    det = net._detectors[key]
    det.max_cand_saved, det.max_cand = det.max_cand, 4        # pretend the candidate buffer was tiny
    try:
        again = detect(net, x, 80, 0.005, 0.45, is_eval=True)
    finally:
        det.max_cand = det.max_cand_saved
    for a, b in zip(again, two_phase):
        assert torch.equal(a, b)
    ids = {k: id(v) for k, v in net._detectors.items()}
    detect(net, x[:2], 80, 0.5, 0.4); detect(net, x, 80, 0.5, 0.4); detect(net, x[:2], 80, 0.5, 0.4)
    assert len(net._detectors) <= dmod.DETECTOR_CACHE_MAX
    detect(net, x, 80, 0.5, 0.4)
    assert sum(1 for k, v in net._detectors.items() if ids.get(k) in (None, id(v))) == len(net._detectors)


def _trained_weights_path():
    cands = [os.environ.get("YV3_TRAINED_WEIGHTS"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weights", "yolov3.weights"),
             os.path.join(os.getcwd(), "weights", "yolov3.weights")]
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.skipif(_trained_weights_path() is None, reason="no trained darknet weight file (YV3_TRAINED_WEIGHTS / weights/yolov3.weights): "
                    "yolov3.weights is not available offline (reference README.md:24-30)")
def test_trained_weights_modes_agree(golden_dir):
    """Only when a TRAINED weight file is present: the default fp16-plane mode has never seen one (every fixture uses synthetic
    weights).  The three fp32-class modes must agree with each other within 1e-4 on the dog image (exact-fp32 MFMA without its
    Winograd stage as the yardstick), give the same candidate set, and the default mode must not report saturation."""
    from yolo_v3_amd import YoloNet
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    x = _input(g, "dog416").cuda()
    net = YoloNet((416, 416)).eval()
    net.loadWeight(_trained_weights_path(), "darknet")
    net = net.cuda()
    out = {}
    for mode in (_ffi.F32, _ffi.F32X3, _ffi.F32H2):
        net.math_mode = mode
        with torch.no_grad():
            out[mode] = net.forward_cat(x).cpu()              # (raises Yv3Error if an fp16 plane saturated)
    for mode in (_ffi.F32X3, _ffi.F32H2):
        e = assert_close_rel(out[mode], out[_ffi.F32], TOL, "trained weights: mode %d vs exact fp32" % mode)
        print("trained weights: mode %d vs exact fp32: %.3g" % (mode, e))
    sc = {m: (d[..., 5:] * d[..., 4:5]).amax(-1) > 0.5 for m, d in out.items()}
    assert torch.equal(sc[_ffi.F32H2], sc[_ffi.F32]) and int(sc[_ffi.F32].sum()) > 0


def test_postprocessing_cache_is_per_stream_and_clearable():
    """`postprocessing()` keeps its candidate / count / output / workspace buffers for repeated calls; calls that may
    overlap (another stream) must not share them, and the cache can be dropped."""
    from yolo_v3_amd import utils as yu, clear_postproc_cache
    clear_postproc_cache()
    g = torch.Generator().manual_seed(11)
    d = torch.rand(2, 300, 9, generator=g)
    d[..., 0:2] = d[..., 0:2] * 100 + 50
    d[..., 2:4] = d[..., 2:4] * 60 + 20
    dg = d.cuda()
    want = oc.postprocess(d, 4, 0.3, 0.4)
    r0 = postprocessing(dg, 4, 0.3, 0.4)
    assert len(yu._PP_CACHE) == 1
    postprocessing(dg, 4, 0.3, 0.4)
    assert len(yu._PP_CACHE) == 1                                   # same stream, same shape: reused
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        r1 = postprocessing(dg, 4, 0.3, 0.4)
    assert len(yu._PP_CACHE) == 2                                   # another stream: its own buffers
    for a, b, c in zip(r0, r1, want):
        assert torch.equal(a, c) and torch.equal(b, c)
    assert all(pp.bytes_allocated() > 0 for pp in yu._PP_CACHE.values())
    clear_postproc_cache()
    assert len(yu._PP_CACHE) == 0


@pytest.mark.parametrize("lb", [True, False])
def test_generate_results_file_matches_manual_composition(tmp_path, lb):
    """reference evaluate.py:208-219 (``generate_results_file``) end to end: image files listed in a text file -> the evaluation
    pipeline's input preparation on the GPU (IaaLetterbox / iaa.Scale + ToTensor) -> eval-mode detection at the reference's 0.005 /
    0.45 -> COCO-results file.  The file equals, byte for byte, the one the writer produces from batches prepared by the ORACLE's
    restatement of that input preparation and pushed through ``predict_and_process`` by hand (SW-eval weights)."""
    from PIL import Image
    from yolo_v3_amd import evaluate
    net = load_sw1_net(synth.eval_weight_stream()).cuda()
    shapes = [(333, 500), (480, 640), (415, 833), (416, 416), (300, 200)]
    paths = []
    for i, (h, w) in enumerate(shapes):
        im = (synth.images(1, 416, 800 + i)[0].transpose(1, 2, 0) * 255).astype(np.uint8)
        im = np.ascontiguousarray(np.tile(im, (3, 3, 1))[:h, :w])
        p = str(tmp_path / ("COCO_val2014_%012d.png" % (100 + i)))
        Image.fromarray(im).save(p)                                         # PNG: lossless, decoder-independent
        paths.append(p)
    txt = str(tmp_path / "list.txt")
    open(txt, "w").write("\n".join(paths) + "\n")
    out = str(tmp_path / "results.json")
    n = evaluate.generate_results_file(net, txt, ["c%d" % k for k in range(80)], out, 2, (416, 416), is_letterbox=lb)
    js = json.load(open(out))
    assert n == len(js) > 0 and {e["image_id"] for e in js} <= set(range(100, 105))
    # manual composition with oracle-prepared inputs
    want = str(tmp_path / "want.json")
    prep = oc.iaa_letterbox_image if lb else oc.iaa_scale_image
    with evaluate.open_json_pred_writer(want, None, lb) as wr:
        for i in range(0, len(paths), 2):
            imgs = [evaluate.read_image_rgb(p) for p in paths[i:i + 2]]
            sample = {"img": torch.stack([prep(im, (416, 416)) for im in imgs]), "org_img": imgs, "img_path": paths[i:i + 2]}
            evaluate.predict_and_process([sample], net, 80, wr)
    assert open(out).read() == open(want).read()
    for e in js:
        h, w = shapes[e["image_id"] - 100]
        assert -1e-3 <= e["bbox"][0] <= w and -1e-3 <= e["bbox"][1] <= h
