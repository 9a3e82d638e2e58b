"""CPU numerics gate for a Winograd F(2x2,3x3) variant of the default (fp16 hi+lo plane) convolution mode.

Emulates, per layer, (a) the shipped direct scheme  w_lo*x_hi + w_hi*x_lo + w_hi*x_hi  (fp32 accumulate) and (b) the
Winograd scheme: V = B^T d B in fp32 (x 1/4, exact, folded back into alpha), U = G g G^T in fp64 -> fp32 with the per-output-
channel power-of-two scaling of the direct scheme, both split hi/lo AFTER the transform, the same three products per
(position, cin) accumulated in fp32, Y = A^T M A in fp32 -- and compares both with an fp64 direct convolution, on
SW-1-like and on hostile operands.  No GPU.   python tools/winograd_numerics.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def split16(v):
    hi = v.clamp(-65504, 65504).half().float()
    lo = (v - hi).half().float()
    return hi, lo


def row_scale(w2d):
    """per-output-channel power-of-two scale bringing max|row| into [1,2) (engine.pack_conv)"""
    m = w2d.abs().amax(dim=1)
    e = torch.where(m > 0, -torch.floor(torch.log2(m.clamp(min=1e-38))), torch.zeros_like(m))
    return torch.exp2(e)


def direct_h2(x, w):
    """shipped scheme: x [B,C,H,W] fp32, w [O,C,3,3] fp32 -> fp32 conv (pad 1)"""
    O = w.shape[0]
    sc = row_scale(w.reshape(O, -1))
    ws = w * sc.view(-1, 1, 1, 1)
    wh, wl = split16(ws)
    xh, xl = split16(x)
    y = F.conv2d(xh, wl, padding=1) + F.conv2d(xl, wh, padding=1) + F.conv2d(xh, wh, padding=1)
    return y / sc.view(1, -1, 1, 1)


def winograd_h2(x, w, vscale=0.25):
    B, C, H, W = x.shape
    O = w.shape[0]
    # weights: U[xi][O][C], fp64 transform -> fp32, scaled per output channel over ALL 16 positions
    U = torch.einsum("ij,ocjk,lk->iloc", G, w.double(), G).float()            # [4,4,O,C]
    sc = row_scale(U.permute(2, 0, 1, 3).reshape(O, -1))
    U = U * sc.view(1, 1, O, 1)
    Uh, Ul = split16(U)
    # input tiles: pad 1, 4x4 windows at stride 2
    xp = F.pad(x, (1, 1, 1, 1))
    th, tw = H // 2, W // 2
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                     # [B,C,th,tw,4,4]
    bt = BT.float()
    V = torch.einsum("ij,bcyxjk,lk->ilbcyx", bt, d, bt) * vscale               # fp32 adds/subs (exact scale)
    Vh, Vl = split16(V)
    M = (torch.einsum("iloc,ilbcyx->ilboyx", Ul, Vh) + torch.einsum("iloc,ilbcyx->ilboyx", Uh, Vl)
         + torch.einsum("iloc,ilbcyx->ilboyx", Uh, Vh))                         # fp32 accumulate over C per position
    at = AT.float()
    Y = torch.einsum("pi,ilboyx,ql->boypxq", at, M, at)                         # [B,O,th,2,tw,2]
    return Y.reshape(B, O, H, W) / (sc.view(1, -1, 1, 1) * vscale)


def report(tag, x, w):
    ref = F.conv2d(x.double(), w.double(), padding=1)
    cabs = F.conv2d(x.double().abs(), w.double().abs(), padding=1)
    K = w.shape[1] * 9
    out = {}
    for name, fn in (("fp32 direct (torch CPU)", lambda: F.conv2d(x, w, padding=1)), ("direct hi/lo (shipped)", lambda: direct_h2(x, w)),
                     ("winograd hi/lo", lambda: winograd_h2(x, w))):
        y = fn().double()
        e_rel = ((y - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
        e_unit = ((y - ref).abs() / (2.0 ** -24 * cabs.clamp(min=1e-300))).max().item()
        out[name] = (e_rel, e_unit)
        print("%-28s %-26s max |d|/max(1,|ref|) %.3g   max |d| / (2^-24 sum|w||x|) %.3g  (sqrt(K)=%.0f)" % (tag, name, e_rel, e_unit, K ** 0.5))
    return out


if __name__ == "__main__":
    torch.set_num_threads(8)
    for cin, cout, H in ((256, 512, 26), (512, 1024, 13)):
        # SW-1-like: W ~ U(-a,a), a = sqrt(6/(1.01 fan_in)); activations post-LeakyReLU O(1..25)
        a = (6.0 / (1.01 * cin * 9)) ** 0.5
        w = (torch.rand(cout, cin, 3, 3) * 2 - 1) * a
        x = F.leaky_relu(torch.randn(2, cin, H + (H % 2), H + (H % 2)) * 4.0, 0.1)
        report("SW-1-like %d->%d @%d" % (cin, cout, H), x, w)
        # hostile (tests/test_gpu_configs.py::test_hostile_conv_level): 6 decades of weights x 3 per filter, inputs to 6e4
        rng = np.random.default_rng(cin)
        wh = rng.choice([-1.0, 1.0], size=(cout, cin, 3, 3)) * 10.0 ** rng.uniform(-6, 0, size=(cout, cin, 3, 3)) * 10.0 ** rng.uniform(-2, 1, size=(cout, 1, 1, 1))
        xh = rng.choice([-1.0, 1.0], size=(2, cin, 12, 12)) * 10.0 ** rng.uniform(-6, np.log10(6e4), size=(2, cin, 12, 12))
        report("hostile %d->%d" % (cin, cout), torch.from_numpy(xh.astype(np.float32)), torch.from_numpy(wh.astype(np.float32)))
