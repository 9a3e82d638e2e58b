"""Pin the CPU oracle (oracle/oracle_cpu.py) against outputs of the reference itself.

The fixtures in tests/golden/ were produced by oracle/make_golden.py, which imports
/root/reference in the build container.  These tests run anywhere (no GPU, no reference).
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle_cpu as oc
from yolo_v3_amd import synth

ANCHORS = oc.DEFAULT_ANCHORS


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_stream_slicing_matches_reference_loader(golden_dir, sw1_stream):
    g = json.load(open(os.path.join(golden_dir, "weights_roundtrip.json")))
    assert sha(sw1_stream) == g["stream_sha256"]            # generator is bit-reproducible
    sd, ptr = oc.state_dict_from_stream(sw1_stream)
    assert ptr == g["ptr"] == 62001757
    assert g["n_convs"] == 75 and len(oc.conv_prefixes()) == 75
    for k, h in g["sha256"].items():
        assert sha(sd[k].numpy()) == h, k
        assert list(sd[k].shape) == g["shapes"][k]


def test_decode_bitwise_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "decode.npz"))
    for key in ("h13_s416", "h26_s416", "h52_s416", "h19_s608", "h76_s608"):
        h, size, step, seed, *mask = [int(v) for v in g[key + "_cfg"]]
        logits = synth.uniform(seed, 7, 2 * 255 * h * h, -6.0, 6.0).reshape(2, 255, h, h)
        out = oc.decode(torch.from_numpy(logits.copy()), ANCHORS, mask, (size, size)).numpy()
        assert np.array_equal(out[:, ::step], g[key + "_out"]), key
        assert sha(out) == bytes(g[key + "_sha"]).hex(), key


def test_iou_bitwise_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "iou.npz"))
    xyxy, b2, cxcywh = (torch.from_numpy(g[k]) for k in ("xyxy", "b2", "cxcywh"))
    assert np.array_equal(oc.iou_matrix(xyxy).numpy(), g["iou_vec"], equal_nan=True)
    assert np.array_equal(oc.bbox_iou(xyxy, b2).numpy(), g["bbox_iou_xyxy"], equal_nan=True)
    assert np.array_equal(oc.bbox_iou(cxcywh, cxcywh[:40], "cxcywh").numpy(), g["bbox_iou_cxcywh"], equal_nan=True)
    assert np.array_equal(oc.cxcywh_to_x1y1x2y2(cxcywh).numpy(), g["to_xyxy"])


def _check_result_list(res, g, name):
    n = int(g[name + "_islist"][0])
    assert len(res) == n, name
    for i, r in enumerate(res):
        exp = g["%s_out%d" % (name, i)]
        assert tuple(r.shape) == tuple(exp.shape), (name, i, r.shape, exp.shape)
        assert np.array_equal(r.numpy(), exp), (name, i)


def test_postprocess_bitwise_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "postproc.npz"))
    for name in [str(n) for n in g["names"]]:
        ct, nt, ev, nms = g[name + "_cfg"]
        res = oc.postprocess(torch.from_numpy(g[name + "_in"].copy()), 80, float(ct), float(nt), bool(ev), bool(nms))
        _check_result_list(res, g, name)


@pytest.mark.parametrize("name", ["dog416", "u416", "u608"])
def test_end_to_end_vs_reference(golden_dir, sw1_stream, name):
    """Same torch CPU ops on the same weights: the functional restatement reproduces the
    reference's detections and final boxes bitwise."""
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    sd, _ = oc.state_dict_from_stream(sw1_stream)
    if name == "dog416":
        x = torch.from_numpy(g["dog_u8"].astype(np.float32) / np.float32(255.0)).permute(2, 0, 1).unsqueeze(0).contiguous()
    else:
        b, s = {"u416": (2, 416), "u608": (1, 608)}[name]
        x = torch.from_numpy(synth.images(b, s, int(g[name + "_seed"][0])))
    with torch.no_grad():
        dets = torch.cat(oc.yolonet_forward(sd, x), 1)
    rows = g[name + "_rows"]
    np.testing.assert_allclose(dets[:, rows].numpy(), g[name + "_dets_rows"], rtol=2e-6, atol=2e-6)
    res = oc.postprocess(dets, 80, 0.5, 0.4)
    assert len(res) == int(g[name + "_nres"][0])
    for i, r in enumerate(res):
        exp = g["%s_boxes%d" % (name, i)]
        assert tuple(r.shape) == tuple(exp.shape)
        np.testing.assert_allclose(r.numpy(), exp, rtol=2e-6, atol=2e-6)
    if name == "dog416":
        ct, nt = g["dog416_eval_cfg"]
        ev = oc.postprocess(dets, 80, float(ct), float(nt), True, True)
        assert tuple(ev[0].shape) == tuple(g["dog416_eval_boxes0"].shape)
        np.testing.assert_allclose(ev[0].numpy(), g["dog416_eval_boxes0"], rtol=2e-6, atol=2e-6)


def test_neighbour_rows_bitwise_vs_reference(golden_dir):
    """correct_yolo_boxes (both modes) and letterbox_transforms against the reference's outputs."""
    g = np.load(os.path.join(golden_dir, "neighbours.npz"))
    boxes = torch.from_numpy(g["boxes"])
    for ci, (ow, oh, iw, ih) in enumerate(g["cases"].tolist()):
        for lb in (0, 1):
            out = oc.correct_yolo_boxes(boxes, ow, oh, iw, ih, bool(lb))
            assert np.array_equal(out.numpy(), g["out_%d_%d" % (ci, lb)]), (ci, lb)
        assert np.array_equal(np.array(oc.letterbox_transforms((ow, oh), (iw, ih)), dtype=np.float64), g["trans_%d" % ci])


def test_bf16_restatement_vs_hooked_reference(golden_dir, sw1_stream):
    """BASELINE configs[2] ("bf16 convs / fp32 decode"): oracle_cpu's ``prec="bf16"`` restatement equals, BIT FOR BIT,
    what the reference's own modules produce when their conv operands are rounded to bfloat16 by forward hooks
    (oracle/make_golden_bf16.py -> tests/golden/e2e_bf16.npz)."""
    g = np.load(os.path.join(golden_dir, "e2e_bf16.npz"))
    sd, _ = oc.state_dict_from_stream(sw1_stream)
    for name in ("u416", "u608"):
        B, size, seed = [int(v) for v in g[name + "_cfg"]]
        x = torch.from_numpy(synth.images(B, size, seed))
        with torch.no_grad():
            d = torch.cat(oc.yolonet_forward(sd, x, prec="bf16"), 1)
        assert np.array_equal(d[:, g[name + "_rows"]].numpy(), g[name + "_dets_rows"]), name
        s, m = g[name + "_dets_sum"]
        assert abs(float(d.double().sum()) - s) <= 1e-9 * abs(s) and float(d.double().abs().max()) == m
        # and it is a different computation from the fp32 path (mean |d|/max(1,|ref|) ~ 1.4e-3)
        with torch.no_grad():
            f = torch.cat(oc.yolonet_forward(sd, x), 1)
        e = ((d - f).abs() / f.abs().clamp(min=1.0)).mean()
        assert 2e-4 < float(e) < 1e-2


def test_boxes_delta_metric():
    """oracle/boxdelta.py: the 'NMS boxes delta vs ref' figure (set-wise pairing within image and class)."""
    from oracle.boxdelta import boxes_delta
    a = [torch.tensor([[0, 0, 10, 10, .9, .8, 3.], [5, 5, 20, 20, .9, .7, 3.], [0, 0, 5, 5, .5, .6, 1.]]), torch.Tensor()]
    b = [torch.tensor([[5, 5, 20, 20.001, .9, .7, 3.], [0, 0, 10, 10, .9, .8001, 3.]]), torch.Tensor()]
    d = boxes_delta(a, b)
    assert (d["images"], d["ref_boxes"], d["got_boxes"], d["matched"], d["unmatched_got"], d["unmatched_ref"]) == (2, 2, 3, 2, 1, 0)
    assert d["count_equal_images"] == 1 and d["class_equal_images"] == 1 and abs(d["unmatched_frac"] - 0.2) < 1e-12
    assert 4e-5 < d["max_rel_err_coords"] < 6e-5 and 0.9e-4 < d["max_abs_err_score"] < 1.1e-4 and d["max_abs_err_conf"] == 0.0
    z = boxes_delta([], [], n_img=3)
    assert z["unmatched_frac"] == 0.0 and z["count_equal_images"] == 3
    same = boxes_delta(a, a)
    assert same["matched"] == 3 and same["max_rel_err_coords"] == 0.0 and same["min_matched_iou"] == 1.0


def test_eval_mode_at_reference_thresholds_vs_reference(golden_dir):
    """evaluate.py:201-204 as the reference runs it (conf 0.005 / nms 0.45 / is_eval=True) with the SW-eval weights:
    the oracle's detections and kept boxes equal the reference's (oracle/make_golden_eval.py)."""
    g = np.load(os.path.join(golden_dir, "e2e_eval.npz"))
    B, size, seed = [int(v) for v in g["in_cfg"]]
    ct, nt = [float(v) for v in g["cfg"]]
    assert (ct, nt) == (0.005, 0.45)
    sd, _ = oc.state_dict_from_stream(synth.eval_weight_stream())
    x = torch.from_numpy(synth.images(B, size, seed))
    with torch.no_grad():
        dets = torch.cat(oc.yolonet_forward(sd, x), 1)
    assert np.abs(dets[:, g["rows"]].numpy() - g["dets_rows"]).max() <= 2e-6 * max(1.0, float(np.abs(g["dets_rows"]).max()))
    sc = dets[..., 5:] * dets[..., 4:5]
    assert (sc > ct).sum((1, 2)).tolist() == g["n_pairs"].tolist() and 1000 <= int(g["n_pairs"].min()) <= 3000
    res = oc.postprocess(dets, 80, ct, nt, True, True)
    for i, r in enumerate(res):
        exp = g["boxes%d" % i]
        assert tuple(r.shape) == tuple(exp.shape)
        assert np.abs(r.numpy() - exp).max() <= 1e-4
        assert np.array_equal(r.numpy()[:, 6], exp[:, 6])


def test_cv_resize_restatement_known_answers():
    """The oracle's restatement of OpenCV's 8-bit fixed-point resize (cv2 is absent here: parity with cv2 itself stays unpinned,
    DESIGN 6b) must at least reproduce the answers that follow from OpenCV's algorithm without running it: a same-size resize is
    the identity (fractional offset 0 -> coefficients (0, 2048, 0, 0) / (2048, 0)), a constant image stays constant for every
    value and scale (the fixed-point coefficient sets sum to 2048 and the rounding shifts are symmetric), and INTER_LINEAR at
    exactly half size is INTER_AREA's 2x2 mean with +2 >> 2 rounding (resize.cpp re-routes that case)."""
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(oc.cv_resize_cubic_u8(img, 53, 37), img)
    assert np.array_equal(oc.cv_resize_linear_u8(img, 53, 37), img)
    for v in (0, 1, 127, 128, 254, 255):
        c = np.full((20, 30, 3), v, np.uint8)
        for (w, h) in ((47, 33), (13, 9), (30, 45), (416, 277)):
            assert np.unique(oc.cv_resize_cubic_u8(c, w, h)).tolist() == [v]
            assert np.unique(oc.cv_resize_linear_u8(c, w, h)).tolist() == [v]
    img2 = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8).astype(np.int32)
    mean = ((img2[0::2, 0::2] + img2[0::2, 1::2] + img2[1::2, 0::2] + img2[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(oc.cv_resize_linear_u8(img2.astype(np.uint8), 30, 20), mean)
    # letterbox geometry: 602 x 452 -> 416 x 416: a 416 x 312 picture at y = 52 on a 128 background (utils.py:34-56)
    pic = rng.integers(0, 256, (452, 602, 3), dtype=np.uint8)
    lb = oc.letterbox_image(pic, (416, 416)).numpy()                       # float32 [3, 416, 416] = uint8 / 255
    grey = np.float32(128) / np.float32(255)
    assert lb.shape == (3, 416, 416) and np.all(lb[:, :52] == grey) and np.all(lb[:, 364:] == grey)
    box = oc.cv_resize_cubic_u8(pic, 416, 312).astype(np.float32) / np.float32(255)
    assert np.array_equal(lb[:, 52:364], box.transpose(2, 0, 1))


def test_eval_letterbox_geometry_and_totensor_vs_reference(golden_dir):
    """SURVEY 8f-1, eval variant: the oracle's (and the product's) restatement of ``IaaLetterbox._compute_height_width_pad``
    (transforms.py:196-205) equals the REFERENCE's output on 75 (shape, dim) cases -- incl. those where it differs by one pixel from
    ``utils.letterbox_transforms`` -- and ``uint8 -> float32 / 255`` is the reference's ToTensor map bit for bit (oracle/make_golden_eval_letterbox.py)."""
    import os
    from yolo_v3_amd import utils as yu
    g = np.load(os.path.join(golden_dir, "eval_letterbox.npz"))
    differs = 0
    for h, w, dw, dh, rw, rh, xp, yp in g["geometry"].tolist():
        assert oc.iaa_letterbox_params((h, w, 3), dh, dw) == (rw, rh, xp, yp)
        assert yu.iaa_letterbox_params((h, w), dh, dw) == (rw, rh, xp, yp)
        bw, bh, bx, by, _ = oc.letterbox_transforms((w, h), (dw, dh))
        assert (bw, bh) == (rw, rh)
        differs += (bx, by) != (xp, yp)
    assert differs > 0                                    # the two pad rules really are different functions
    ramp = g["ramp"]
    mine = (ramp.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)
    assert mine.dtype == np.float32 and np.array_equal(mine, g["ramp_tensor"])
