"""Shared helpers for the parity tests."""
import numpy as np
import torch

TOL = 1e-4          # north-star tolerance: |d| <= 1e-4 * max(1, |ref|), fp32


def rel_err(a, ref):
    a, ref = torch.as_tensor(a).detach().double(), torch.as_tensor(ref).detach().double()
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


# ----------------------------------------------------------------------------- layer-by-layer ("teacher-forced") runs
def teacher_forced_layers(net, mode, x, taps):
    """Run every one of the 75 convolutions of `net` ALONE through the C-ABI in math mode `mode`, each fed the
    ORACLE's activations (``taps`` = [(name, NCHW fp32 tensor)] from oracle_cpu.head_logits) instead of the previous
    HIP layer's output.  Errors therefore do not compound: what is returned, ``{name: (got NCHW fp32, ref)}``, isolates
    each kernel launch's own arithmetic (summation order, operand splitting, output rounding)."""
    import ctypes
    from yolo_v3_amd import engine as _engine, _ffi
    eng = net.engine(mode)
    eng.ensure_packed()
    old = eng.fuse_decode
    eng.fuse_decode, eng._plans = False, {}                      # materialise the head convs' logits
    out = {}
    try:
        B, _, H, W = x.shape
        with torch.cuda.device(eng.device):
            plan = eng.plan(B, H, W)
            tapd = dict(taps)
            ptr2name = {buf.data_ptr(): name for name, buf in plan.layer_out.items()}

            def put(ptr):
                if not ptr:
                    return
                name = ptr2name[ptr]
                buf = plan.layer_out[name]
                buf.copy_(_engine.to_planes(tapd[name].permute(0, 2, 3, 1).contiguous().cuda(), mode))

            def get(name):
                return _engine.from_planes(plan.layer_out[name], mode).permute(0, 3, 1, 2).float().cpu()

            eng.run_conv0(plan, eng.prepare_input(x.cuda()))
            out[eng.specs[0].name] = (get(eng.specs[0].name), tapd[eng.specs[0].name])
            for j in range(plan.n_desc):
                d = plan.descs[j]
                name = eng.specs[plan.desc_spec[j]].name          # (a layer may be two launches over batch slices)
                for ptr in (d.x, d.x2, d.residual):
                    put(ptr)
                _ffi.check(_ffi.lib().yv3_conv2d(ctypes.byref(d), _ffi.stream_ptr()), "yv3_conv2d " + name)
                out[name] = (get(name), tapd[name])
            torch.cuda.synchronize()
            eng.raise_if_overflowed(plan, int(plan.flags.item()))
    finally:
        eng.fuse_decode, eng._plans = old, {}
    return out


def bf16_ulp(t):
    """Spacing of bfloat16 numbers at the magnitude of each element of fp32 tensor `t` (>= the smallest normal's)."""
    e = torch.floor(torch.log2(t.abs().double().clamp(min=2.0 ** -126)))
    return torch.exp2(e - 7)


# ----------------------------------------------------------------------------- hostile synthetic network
def hostile_state_dict(seed=7, size=416, num_class=80, gamma_decades=(-5.0, 2.5), res_gamma_hi=1.5):
    """A YOLOv3 parameter set built to stress the fp16-plane arithmetic of the default math mode, calibrated with
    the CPU oracle so that it is still a WORKING network (BN statistics are the statistics of the data that
    actually reaches each BatchNorm, as in a trained model):

      * conv weights: random sign x log-uniform magnitude over 6 decades inside every filter, times a per-output-
        channel factor 10^U(-2,1.5) (rows whose weights are all tiny relative to the layer maximum);
      * BN running_var / running_mean: measured on the calibration image -> variances spread over >= 7 decades;
      * BN gamma: log-uniform 1e-5 .. 3e2 (capped for residual branches), beta = U(-0.2,0.2)*gamma: activations
        spread from ~1e-6 to ~1e4 inside one tensor;
      * head convs scaled so the logits have unit-order spread.

    Returns (state_dict for oracle_cpu, calibration image [1,3,size,size]).  Deterministic (numpy Generator)."""
    import torch.nn.functional as F
    from oracle import oracle_cpu as oc
    from yolo_v3_amd import synth
    rng = np.random.default_rng(seed)
    sd = {}
    x = torch.from_numpy(synth.images(1, size, 900 + seed))
    for prefix, cin, cout, k, has_bn in oc.conv_prefixes(num_class):
        mag = 10.0 ** rng.uniform(-6.0, 0.0, size=(cout, cin, k, k))
        sgn = rng.choice([-1.0, 1.0], size=(cout, cin, k, k))
        row = 10.0 ** rng.uniform(-2.0, 1.5, size=(cout, 1, 1, 1)) if has_bn else np.ones((cout, 1, 1, 1))
        w = torch.from_numpy((mag * sgn * row).astype(np.float32))
        if has_bn:
            sd[prefix + ".conv.weight"] = w
            res2 = prefix.endswith(".conv2")
            g = 10.0 ** rng.uniform(gamma_decades[0], res_gamma_hi if res2 else gamma_decades[1], size=cout)
            sd[prefix + ".bn.weight"] = torch.from_numpy(g.astype(np.float32))
            sd[prefix + ".bn.bias"] = torch.from_numpy((rng.uniform(-0.2, 0.2, size=cout) * g).astype(np.float32))
        else:
            sd[prefix + ".weight"] = w
            sd[prefix + ".bias"] = torch.from_numpy(rng.uniform(-1.0, 1.0, size=cout).astype(np.float32))
    orig_cbr = oc.cbr

    def calibrating_cbr(sd_, prefix, xin, stride=1, prec=None, store=True):
        w = sd_[prefix + ".conv.weight"]
        y = F.conv2d(xin, w, None, stride, (w.shape[2] - 1) // 2)
        # normalise the filter bank so the median channel variance is 1, then record the measured statistics
        v = y.var(dim=(0, 2, 3), unbiased=False)
        s = float(1.0 / v.median().clamp(min=1e-30).sqrt())
        sd_[prefix + ".conv.weight"] = w * s
        y = y * s
        sd_[prefix + ".bn.running_mean"] = y.mean(dim=(0, 2, 3)).float()
        sd_[prefix + ".bn.running_var"] = y.var(dim=(0, 2, 3), unbiased=False).float().clamp(min=1e-12)
        return orig_cbr(sd_, prefix, xin, stride, prec, store)

    oc.cbr = calibrating_cbr
    try:
        with torch.no_grad():
            feat, r36, r61 = oc.backbone(sd, x)
            cur, route = feat, None
            for pre, tail in (("pre_det1", None), ("pre_det2", r61), ("pre_det3", r36)):
                if tail is not None:
                    cur = oc.upsample_cat(sd, "up1" if pre == "pre_det2" else "up2", route, tail)
                # head conv: scale to logits of unit-order spread (keeps exp() in range)
                h = cur
                for i in range(6):
                    h = oc.cbr(sd, "%s.mlist.%d" % (pre, i), h)
                    if i == 4:
                        route = h
                lg = F.conv2d(h, sd[pre + ".mlist.6.weight"])
                sd[pre + ".mlist.6.weight"] = sd[pre + ".mlist.6.weight"] * float(1.5 / lg.std().clamp(min=1e-30))
    finally:
        oc.cbr = orig_cbr
    return sd, x


def state_dict_to_stream(sd, num_class=80):
    """oracle state_dict -> darknet float stream (inverse of oracle_cpu.state_dict_from_stream)."""
    from oracle import oracle_cpu as oc
    parts = []
    for prefix, cin, cout, k, has_bn in oc.conv_prefixes(num_class):
        keys = oc._cbr_keys(prefix) if has_bn else [prefix + ".bias", prefix + ".weight"]
        parts += [sd[key].detach().float().numpy().ravel() for key in keys]
    return np.concatenate(parts).astype(np.float32)


def detector_dets(net):
    """The detections tensor of the detector ``detect()`` / ``detect_sharded()`` used last (an LRU cache on the net): the
    "identical detections" for bit-exact post-processing checks.  (``net.forward_cat`` runs the one-lane plan of the whole
    batch; a Detector may run two lanes of half the batch, for which the library may choose another form of a layer --
    Winograd or direct, stream-K or not -- so its detections can differ from forward_cat's in the last bits.)"""
    return list(net._detectors.values())[-1].dets
