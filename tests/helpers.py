"""Shared helpers for the parity tests."""
import numpy as np
import torch

TOL = 1e-4          # north-star tolerance: |d| <= 1e-4 * max(1, |ref|), fp32


def rel_err(a, ref):
    a, ref = torch.as_tensor(a).double(), torch.as_tensor(ref).double()
    return ((a - ref).abs() / ref.abs().clamp(min=1.0))


def assert_close_rel(a, ref, tol=TOL, what=""):
    e = rel_err(a, ref)
    assert torch.isfinite(e).all(), "%s: non-finite difference" % what
    assert float(e.max()) <= tol, "%s: max normalised error %.3g > %.1g" % (what, float(e.max()), tol)
    return float(e.max())


def match_boxes(got, exp, tol=TOL):
    """Order-insensitive comparison of two [n,7] box lists (x1,y1,x2,y2,conf,score,cls): same count,
    same class multiset, and a one-to-one pairing within each class with every column within tol.

    Normalisation: conf/score by max(1,|ref|) (= 1); the four corner coordinates by the box's
    largest coordinate magnitude, max(1, |x1|,|y1|,|x2|,|y2|).  Corners are x = cx -/+ w/2
    (boundingbox.py:25-29): when cx ~ w/2 the subtraction cancels, so the achievable error of a
    corner is tol * the magnitude of its OPERANDS, not of the (possibly tiny) difference.  The
    un-cancelled quantities (cx,cy,w,h,conf,cls) are checked elementwise by assert_close_rel.
    Returns the worst normalised error."""
    got, exp = torch.as_tensor(got).double(), torch.as_tensor(exp).double()
    assert got.numel() == 0 or got.dim() == 2
    assert tuple(got.shape) == tuple(exp.shape), "box count differs: got %s expected %s" % (tuple(got.shape), tuple(exp.shape))
    if got.numel() == 0:
        return 0.0
    worst = 0.0
    for c in exp[:, 6].unique():
        g, e = got[got[:, 6] == c], exp[exp[:, 6] == c]
        assert len(g) == len(e), "class %d: got %d boxes, expected %d" % (int(c), len(g), len(e))
        scale = torch.ones_like(e[:, :6])
        scale[:, :4] = e[:, :4].abs().amax(1, keepdim=True).clamp(min=1.0)
        d = ((g[:, None, :6] - e[None, :, :6]).abs() / scale[None]).amax(-1)                            # [ng, ne]
        used = set()
        for j in range(len(e)):
            i = int(d[:, j].argmin())
            assert i not in used, "class %d: two expected boxes map to the same result" % int(c)
            used.add(i)
            assert float(d[i, j]) <= tol, "class %d box %d: error %.3g > %.1g" % (int(c), j, float(d[i, j]), tol)
            worst = max(worst, float(d[i, j]))
    return worst


def check_result_convention(res, exp):
    """[] sentinel / list length / empty-image shapes exactly like the reference."""
    assert isinstance(res, list)
    assert len(res) == len(exp)
    for r, e in zip(res, exp):
        assert tuple(r.shape) == tuple(e.shape), (tuple(r.shape), tuple(e.shape))


def load_sw1_net(stream, size=416, num_class=80):
    from yolo_v3_amd import YoloNet, WeightManager
    net = YoloNet((size, size), numClass=num_class).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    return net
