"""CPU numerics gate for Winograd F(4x4,3x3) in the EXACT fp32 mode (round 6).

The whole network through the oracle (oracle/oracle_cpu.py) with every eligible layer (3x3, stride 1, cin >= 64: 31 of the
75 convolutions) replaced by an fp32 EMULATION of a Winograd form -- U = G g G^T in fp64 -> fp32 (as engine.pack does),
V = B^T d B in fp32, M = sum_c U V accumulated in fp32, Y = A^T M A in fp32 -- and compared three ways:

  * head logits against an fp64 evaluation of the direct form (the truth), next to the fp32 oracle's own distance;
  * decoded detections against the fp32 oracle (what the parity tests assert, bar 1e-4);
  * final boxes (boxes_delta) against the fp32 oracle's.

Forms: direct (= the oracle), F(2x2), F(4x4) with the SHIPPED points (0, 1, -1, 1/2, -2, inf), with the Lavin points
(0, +-1, +-2, inf) and with (0, +-1, +-1/2, inf), the latter two also restricted to the high-resolution layers.   Data: SW-1 on synthetic scenes (the headline data) and
the hostile calibrated set (tests/helpers.hostile_state_dict).   python tools/winograd_f32_gate.py [n_images]
"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import oracle_cpu as oc
from oracle.boxdelta import boxes_delta
from yolo_v3_amd import synth
from tests.helpers import hostile_state_dict, rel_err


def cook_toom(points, m=4, r=3):
    """Winograd matrices (A^T [m,n], G [n,r], B^T [n,n]) for F(m,r) from n-1 finite points + infinity, in fp64, by the
    Toom-Cook construction (Vandermonde evaluation / Lagrange interpolation), n = m + r - 1."""
    n = m + r - 1
    p = np.array(points, dtype=np.float64)
    assert len(p) == n - 1
    # A^T: evaluation of a degree m-1 polynomial... rows i = x^i at the points, last column = infinity (leading coeff)
    AT = np.zeros((m, n)); G = np.zeros((n, r));
    for j in range(n - 1):
        for i in range(m):
            AT[i, j] = p[j] ** i
        for i in range(r):
            G[j, i] = p[j] ** i
    AT[m - 1, n - 1] = 1.0
    G[n - 1, r - 1] = 1.0
    # scale G rows by 1 / prod_{k != j}(p_j - p_k)
    for j in range(n - 1):
        G[j] /= np.prod([p[j] - p[k] for k in range(n - 1) if k != j])
    # B^T from the identity  y = A^T [(G g) . (B^T d)]  for all g, d: solve by linear algebra.
    # M(x) = prod (x - p_k); row j (finite) of B^T = coefficients of M(x)/(x - p_j); last row = coefficients of M(x).
    Mx = np.poly1d(np.poly(p))                                 # highest power first
    BT = np.zeros((n, n))
    for j in range(n - 1):
        q = np.polydiv(Mx.coeffs, np.array([1.0, -p[j]]))[0]   # degree n-2
        BT[j, :n - 1] = q[::-1]
    BT[n - 1, :] = Mx.coeffs[::-1]
    return AT, G, BT


def check_matrices(AT, G, BT, m=4, r=3):
    rng = np.random.default_rng(0)
    g = rng.standard_normal(r); d = rng.standard_normal(m + r - 1)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(g[k] * d[i + k] for k in range(r)) for i in range(m)])
    assert np.allclose(y, ref, rtol=1e-10, atol=1e-10), (y, ref)


def winograd_conv_f32(x, w, mats, m):
    """x [B,C,H,W] fp32, w [O,C,3,3] fp32, pad 1, stride 1 -> [B,O,H,W] fp32, every step in fp32 but the weight transform."""
    AT, G, BT = (torch.from_numpy(a) for a in mats)
    n = m + 2
    B, C, H, W = x.shape
    O = w.shape[0]
    th, tw = -(-H // m), -(-W // m)
    U = torch.einsum("ij,ocjk,lk->iloc", G, w.double(), G).float()                       # [n,n,O,C]
    xp = F.pad(x, (1, 1 + tw * m - W, 1, 1 + th * m - H))
    d = xp.unfold(2, n, m).unfold(3, n, m)                                                # [B,C,th,tw,n,n]
    bt = BT.float()
    # two 1-D passes in fp32 (as a kernel would: rows, then columns)
    t = torch.einsum("ij,bcyxjk->bcyxik", bt, d)
    V = torch.einsum("bcyxik,lk->ilbyxc", t, bt).contiguous()                             # [n,n,B,th,tw,C]
    M = torch.matmul(V.reshape(n, n, B * th * tw, C), U.transpose(2, 3))                  # fp32 GEMM per position -> [n,n,T,O]
    at = AT.float()
    t = torch.einsum("pi,ilto->plto", at, M)
    Y = torch.einsum("plto,ql->tpqo", t, at)                                              # [T,m,m,O]
    Y = Y.reshape(B, th, tw, m, m, O).permute(0, 5, 1, 3, 2, 4).reshape(B, O, th * m, tw * m)
    return Y[:, :, :H, :W].contiguous()


class FShim(types.SimpleNamespace):
    pass


def patched_forward(sd, x, form):
    """head logits + decoded detections of the oracle with the eligible layers in Winograd form `form` (None = direct)."""
    real = F
    count = [0]

    def conv2d(inp, w, bias=None, stride=1, padding=0, *a, **k):
        if form is not None and w.shape[2] == 3 and stride == 1 and w.shape[1] >= 64 and inp.dtype == torch.float32:
            count[0] += 1
            f = form if (len(form) < 3 or inp.shape[2] >= form[2]) else form[3]          # (mats, m, min_h, fallback form)
            return winograd_conv_f32(inp, w, f[0], f[1])
        return real.conv2d(inp, w, bias, stride, padding, *a, **k)
    shim = FShim(**{k: getattr(real, k) for k in dir(real) if not k.startswith("__")})
    shim.conv2d = conv2d
    old = oc.F
    oc.F = shim
    try:
        with torch.no_grad():
            lg = oc.head_logits(sd, x)
            dets = torch.cat(oc.yolonet_forward(sd, x), 1)
    finally:
        oc.F = old
    return lg, dets, count[0]


def run(tag, sd, x, forms, conf=0.5):
    with torch.no_grad():
        l64 = oc.head_logits({k: v.double() for k, v in sd.items()}, x.double())
    base = None
    for name, form in forms:
        lg, dets, n = patched_forward(sd, x, form)
        e64 = max(float(rel_err(a, b).max()) for a, b in zip(lg, l64))
        if base is None:
            base = (lg, dets, oc.postprocess(dets.clone(), 80, conf, 0.4))
            print("%-10s %-22s logits vs fp64 %.3g   (this IS the fp32 oracle)" % (tag, name, e64), flush=True)
            continue
        e32 = max(float(rel_err(a, b).max()) for a, b in zip(lg, base[0]))
        ok = torch.isfinite(base[1]) & (base[1].abs() < 1e30)
        e_det = float(rel_err(dets[ok], base[1][ok]).max())
        bd = boxes_delta(oc.postprocess(dets.clone(), 80, conf, 0.4), base[2], n_img=x.shape[0])
        print("%-10s %-22s logits vs fp64 %.3g  vs fp32 oracle %.3g | detections vs oracle %.3g | boxes %d/%d matched, coords %.3g conf %.3g score %.3g  [%d layers]"
              % (tag, name, e64, e32, e_det, bd["matched"], bd["ref_boxes"], bd["max_rel_err_coords"], bd["max_abs_err_conf"], bd["max_abs_err_score"], n), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    f2 = cook_toom([0, 1, -1], m=2); check_matrices(*f2, m=2)
    f4 = cook_toom([0, 1, -1, 2, -2]); check_matrices(*f4)
    f4h = cook_toom([0, 1, -1, 0.5, -0.5]); check_matrices(*f4h)
    f4s = cook_toom([0, 1, -1, 0.5, -2]); check_matrices(*f4s)          # the SHIPPED points (csrc/conv_wino4_f32.hip)
    forms = [("direct", None), ("F(2x2)", (f2, 2)), ("F(4x4) 0,1,-1,1/2,-2 SHIPPED", (f4s, 4)), ("F(4x4) 0,+-1,+-2", (f4, 4)), ("F(4x4) 0,+-1,+-1/2", (f4h, 4)),
             ("F(4x4) H>=48, else F(2x2)", (f4, 4, 48, (f2, 2))), ("F(4x4) H>=24, else F(2x2)", (f4, 4, 24, (f2, 2))),
             ("F(4x4)+-1/2 H>=48, else F2", (f4h, 4, 48, (f2, 2))), ("F(4x4)+-1/2 H>=24, else F2", (f4h, 4, 24, (f2, 2)))]
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.stream_to_state_dict(synth.weight_stream()).items()}
    x = torch.from_numpy(synth.images(nimg, 416, 0))
    run("SW-1 416", sd, x, forms)
    x = torch.from_numpy(synth.images(2, 608, 4))
    run("SW-1 608", sd, x, forms)
    sdh, xh = hostile_state_dict()
    run("hostile", sdh, xh, forms)
