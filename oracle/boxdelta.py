""""NMS boxes delta vs ref" (BASELINE.json metric)  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

Compares two results of the reference's caller idiom (test.py:35-36: a list of per-image ``[n,7]`` tensors
``x1,y1,x2,y2,conf,score,cls``, or ``[]``) set-wise, the way SURVEY.md 8(d) prescribes for inputs whose
decisions may sit inside fp32 noise of a threshold: boxes are paired one-to-one within (image, class) by best
IOU, a pair counts as matched when IOU >= ``iou_match`` (0.999), and the numeric deltas are taken over the
matched pairs.  Like the rest of ``oracle/`` it is imported only by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg.
"""
import numpy as np
import torch


def _iou_pairs(a, b):
    """[na,nb] IOU of x1y1x2y2 boxes in float64 (reference utils.py:122-146 formula)."""
    ix1 = np.maximum(a[:, None, 0], b[None, :, 0]); iy1 = np.maximum(a[:, None, 1], b[None, :, 1])
    ix2 = np.minimum(a[:, None, 2], b[None, :, 2]); iy2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(ix2 - ix1, 0, None) * np.clip(iy2 - iy1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (aa[:, None] + ab[None, :] - inter)


def _as_list(res, n_img):
    if isinstance(res, list) and len(res) == 0:            # the reference's "nothing anywhere" sentinel (utils.py:248)
        return [np.zeros((0, 7))] * n_img
    out = []
    for r in res:
        r = torch.as_tensor(r).double().numpy() if not isinstance(r, np.ndarray) else r.astype(np.float64)
        out.append(r.reshape(-1, 7) if r.size else np.zeros((0, 7)))
    return out


def boxes_delta(got, ref, n_img=None, iou_match=0.999):
    """Set-wise difference of two detection results.  Returns a dict:

    images, ref_boxes, got_boxes, matched            counts
    unmatched_ref, unmatched_got, unmatched_frac      boxes without a partner / (ref + got boxes)
    count_equal_images                               images with the same number of boxes in both results
    class_equal_images                               images whose per-class box counts agree exactly (pairs are formed
                                                     within a class, so every matched pair has equal class ids)
    max_rel_err_coords   over matched pairs, |d(x1,y1,x2,y2)| / max(1, largest |coordinate| of the ref box)
    max_abs_err_conf, max_abs_err_score              over matched pairs (both are <= 1, so abs == rel to max(1,|ref|))
    min_matched_iou
    """
    n_img = n_img if n_img is not None else max(len(got), len(ref))
    G, R = _as_list(got, n_img), _as_list(ref, n_img)
    assert len(G) == len(R) == n_img, (len(G), len(R), n_img)
    out = dict(images=n_img, ref_boxes=0, got_boxes=0, matched=0, unmatched_ref=0, unmatched_got=0,
               count_equal_images=0, class_equal_images=0, max_rel_err_coords=0.0, max_abs_err_conf=0.0,
               max_abs_err_score=0.0, min_matched_iou=1.0)
    for g, r in zip(G, R):
        out["ref_boxes"] += len(r); out["got_boxes"] += len(g)
        same_counts = True
        for c in np.union1d(g[:, 6], r[:, 6]):
            gc, rc = g[g[:, 6] == c], r[r[:, 6] == c]
            if len(gc) != len(rc):
                same_counts = False
            if len(gc) == 0 or len(rc) == 0:
                out["unmatched_ref"] += len(rc); out["unmatched_got"] += len(gc)
                continue
            iou = np.nan_to_num(_iou_pairs(rc[:, :4], gc[:, :4]), nan=0.0)
            used_g = np.zeros(len(gc), dtype=bool)
            n_match = 0
            # best pairs first (greedy on IOU): exact for the near-identical sets this is meant for
            ci, cj = np.nonzero(iou >= iou_match)
            order = np.argsort(-iou[ci, cj], kind="stable")
            used_r = np.zeros(len(rc), dtype=bool)
            for i, j in zip(ci[order], cj[order]):
                if used_r[i] or used_g[j]:
                    continue
                used_r[i] = used_g[j] = True
                n_match += 1
                scale = max(1.0, float(np.abs(rc[i, :4]).max()))
                out["max_rel_err_coords"] = max(out["max_rel_err_coords"], float(np.abs(gc[j, :4] - rc[i, :4]).max()) / scale)
                out["max_abs_err_conf"] = max(out["max_abs_err_conf"], float(abs(gc[j, 4] - rc[i, 4])))
                out["max_abs_err_score"] = max(out["max_abs_err_score"], float(abs(gc[j, 5] - rc[i, 5])))
                out["min_matched_iou"] = min(out["min_matched_iou"], float(iou[i, j]))
            out["matched"] += n_match
            out["unmatched_ref"] += len(rc) - n_match
            out["unmatched_got"] += len(gc) - n_match
        out["count_equal_images"] += int(len(g) == len(r))
        out["class_equal_images"] += int(same_counts)
    tot = out["ref_boxes"] + out["got_boxes"]
    out["unmatched_frac"] = (out["unmatched_ref"] + out["unmatched_got"]) / tot if tot else 0.0
    return out
