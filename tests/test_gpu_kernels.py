"""GPU parity tests for the individual HIP kernels (run with -m gpu on the MI355X).

Every test calls through the C-ABI (ctypes) -- either directly or via the drop-in Python
surface -- and checks against the committed golden fixtures (outputs of the reference) or the
CPU oracle on the same seeded inputs.  Integer/compare work is bit-exact; floating point is
within the tolerance written in each test.
"""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_cpu as oc
from yolo_v3_amd import _ffi, arch, synth, engine
from yolo_v3_amd import YoloLayer, postprocessing, iou_vectorized, bbox_iou, bbox_cxcywh_to_x1y1x2y2
from yolo_v3_amd.darknet import conv_bn_relu, res_layer, UpsampleGroup, PreDetectionConvGroup
from tests.helpers import assert_close_rel, check_result_convention, rel_err

pytestmark = pytest.mark.gpu
ANCHOR_PAIRS = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    assert os.path.exists(_ffi.LIB_PATH), "libyv3.so missing on a GPU box"
    torch.cuda.set_device(0)


# ----------------------------------------------------------------------------- decode
@pytest.mark.parametrize("key", ["h13_s416", "h26_s416", "h52_s416", "h19_s608", "h76_s608"])
def test_decode_vs_reference_golden(golden_dir, key):
    """yololayer.py:31-59,97-105.  Same logits, same op order; only the rounding of exp / sigmoid may differ (csrc/yv3_common.h:
    one v_exp_f32 with a compensated argument + one v_rcp_f32 per element, <= 4 ulp): rtol 1e-6 (fp32), plus
    1e-7 abs for values near 0."""
    g = np.load(os.path.join(golden_dir, "decode.npz"))
    h, size, step, seed, *mask = [int(v) for v in g[key + "_cfg"]]
    logits = synth.uniform(seed, 7, 2 * 255 * h * h, -6.0, 6.0).reshape(2, 255, h, h)
    x = torch.from_numpy(logits.copy()).cuda()
    out = YoloLayer(ANCHOR_PAIRS, mask, (size, size), 80)(x, (size, size)).cpu().numpy()
    assert out.shape == (2, h * h * 3, 85)
    np.testing.assert_allclose(out[:, ::step], g[key + "_out"], rtol=1e-6, atol=1e-7)
    # NHWC entry (the one YoloNet uses): identical arithmetic -> bitwise equal to the NCHW entry
    nhwc = x.permute(0, 2, 3, 1).contiguous()
    out2 = torch.empty(2, h * h * 3, 85, device="cuda")
    flat = []
    for m in mask:
        flat += [float(ANCHOR_PAIRS[m][0]), float(ANCHOR_PAIRS[m][1])]
    _ffi.check(_ffi.lib().yv3_decode(nhwc.data_ptr(), 255, (ctypes.c_float * 6)(*flat), size / h, out2.data_ptr(),
                                     h * h * 3 * 85, 2, h, h, 80, _ffi.stream_ptr()))
    assert np.array_equal(out2.cpu().numpy(), out)


def test_decode_non_square_and_small_classes():
    """Non power-of-two stride (anchor/stride*stride is not exact) and C != 80, against the oracle."""
    torch.manual_seed(3)
    x = torch.randn(2, 3 * 9, 10, 15) * 2
    ref = oc.decode(x, [a for p in ANCHOR_PAIRS for a in p], (3, 4, 5), (450, 300), num_class=4)   # img_dim = (W, H)
    out = YoloLayer(ANCHOR_PAIRS, [3, 4, 5], (450, 300), 4)(x.cuda(), (450, 300)).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)


def test_decode_non_finite_and_huge_logits_like_the_reference():
    """exp(+inf) = inf, sigmoid(+inf) = 1, sigmoid(-inf) = 0, and the same for |t| ~ 1e38 / 3e38 (where t * log2(e)
    overflows); NaN logits stay NaN -- exactly what the reference's torch.exp / torch.sigmoid return (oracle)."""
    vals = [float("inf"), float("-inf"), 1e38, -1e38, 3e38, -3e38, 89.0, -104.0, 100.0, -110.0, float("nan"), 0.0]
    x = torch.zeros(1, 3 * 9, 4, 3)
    x.view(-1)[:len(vals) * 27:27] = torch.tensor(vals)              # spread over positions ...
    for c in range(27):                                              # ... and put every value into every attribute kind
        x[0, c, c % 4, c % 3] = vals[c % len(vals)]
    ref = oc.decode(x, [a for p in ANCHOR_PAIRS for a in p], (3, 4, 5), (96, 128), num_class=4)
    out = YoloLayer(ANCHOR_PAIRS, [3, 4, 5], (96, 128), 4)(x.cuda(), (96, 128)).cpu()
    assert torch.equal(torch.isnan(out), torch.isnan(ref))
    assert torch.equal(torch.isinf(out), torch.isinf(ref))
    fin = torch.isfinite(ref)
    np.testing.assert_allclose(out[fin].numpy(), ref[fin].numpy(), rtol=1e-6, atol=1e-7)
    assert torch.equal(out[torch.isinf(ref)], ref[torch.isinf(ref)])


# ----------------------------------------------------------------------------- geometry
def test_iou_and_box_conversion_bit_exact(golden_dir):
    """utils.py:98-146, boundingbox.py:25-29: plain IEEE fp32 ops -> bitwise equal to the reference."""
    g = np.load(os.path.join(golden_dir, "iou.npz"))
    xyxy, b2, cxcywh = (torch.from_numpy(g[k]).cuda() for k in ("xyxy", "b2", "cxcywh"))
    assert np.array_equal(iou_vectorized(xyxy).cpu().numpy(), g["iou_vec"], equal_nan=True)
    assert np.array_equal(bbox_iou(xyxy, b2).cpu().numpy(), g["bbox_iou_xyxy"], equal_nan=True)
    assert np.array_equal(bbox_iou(cxcywh, cxcywh[:40], mode="cxcywh").cpu().numpy(), g["bbox_iou_cxcywh"], equal_nan=True)
    assert np.array_equal(bbox_cxcywh_to_x1y1x2y2(cxcywh.clone()).cpu().numpy(), g["to_xyxy"])
    # CPU tensors are accepted (moved to the GPU and back) like any other caller input
    assert np.array_equal(iou_vectorized(torch.from_numpy(g["xyxy"])).numpy(), g["iou_vec"], equal_nan=True)


# ----------------------------------------------------------------------------- post-processing
def test_postprocessing_bit_exact_vs_reference_golden(golden_dir):
    """utils.py:226-258 on the hand-built cases: [] sentinel, empty images, argmax ties, chains
    A>B>C, zero-area boxes, eval mode, raw mode, nms_thr >= 1 -- outputs equal bit for bit."""
    g = np.load(os.path.join(golden_dir, "postproc.npz"))
    for name in [str(n) for n in g["names"]]:
        ct, nt, ev, nms = g[name + "_cfg"]
        d = torch.from_numpy(g[name + "_in"].copy())
        keep = d.clone()
        res = postprocessing(d.cuda(), 80, float(ct), float(nt), bool(ev), bool(nms))
        n = int(g[name + "_islist"][0])
        assert isinstance(res, list) and len(res) == n, name
        for i, r in enumerate(res):
            exp = g["%s_out%d" % (name, i)]
            assert not r.is_cuda
            assert tuple(r.shape) == tuple(exp.shape), (name, i, tuple(r.shape), exp.shape)
            assert np.array_equal(r.numpy(), exp), (name, i)
        res_cpu = postprocessing(d, 80, float(ct), float(nt), bool(ev), bool(nms))     # CPU tensor in
        assert len(res_cpu) == n and torch.equal(d, keep), "input must not be modified"


def clustered_detections(B, N, hot):
    """[B, N, 85] detections: ~hot rows per image pass conf 0.5, centres clustered in five spots, six classes."""
    u = synth.uniform01(900 + N, 1, B * N * 8).reshape(B, N, 8)
    d = np.zeros((B, N, 85), dtype=np.float32)
    centres = np.array([[80, 80], [200, 120], [320, 300], [120, 330], [260, 260]], dtype=np.float32)
    k = (u[..., 0] * 5).astype(np.int64)
    d[..., 0] = centres[k][..., 0] + (u[..., 1] - 0.5) * 70
    d[..., 1] = centres[k][..., 1] + (u[..., 2] - 0.5) * 70
    d[..., 2] = 20 + u[..., 3] * 60
    d[..., 3] = 20 + u[..., 4] * 60
    d[..., 4] = np.where(np.arange(N)[None, :] % (N // hot) == 0, 0.55 + 0.44 * u[..., 5], 0.3 * u[..., 5])
    d[..., 5:] = synth.uniform01(901 + N, 2, B * N * 80).reshape(B, N, 80) * 0.6
    cls = (u[..., 6] * 6).astype(np.int64) * 13
    bi, ri = np.meshgrid(np.arange(B), np.arange(N), indexing="ij")
    d[bi, ri, 5 + cls] = 0.7 + 0.29 * u[..., 7]
    return torch.from_numpy(d)


@pytest.mark.parametrize("B,N,hot,is_eval", [(3, 2000, 700, False), (2, 4000, 1500, False), (2, 1500, 300, True)])
def test_postprocessing_dense_vs_oracle(B, N, hot, is_eval):
    """Clustered boxes with heavy suppression (SURVEY App. C-2 recipe), bit-exact vs the oracle."""
    dt = clustered_detections(B, N, hot)
    thr = 0.3 if is_eval else 0.5
    ref = oc.postprocess(dt, 80, thr, 0.4, is_eval, True)
    res = postprocessing(dt.cuda(), 80, thr, 0.4, is_eval, True)
    check_result_convention(res, ref)
    for r, e in zip(res, ref):
        assert torch.equal(r, e)
    assert sum(len(r) for r in res) > 50 * B
    ref = oc.postprocess(dt, 80, thr, 0.4, is_eval, False)
    res = postprocessing(dt.cuda(), 80, thr, 0.4, is_eval, False)
    for r, e in zip(res, ref):
        assert torch.equal(r, e)


def test_candidate_buffer_overflow_is_an_error_not_a_crash():
    """More candidates than the caller-sized buffers hold: the counts say so and to_list raises (no silent truncation); nothing
    is written out of bounds (the NMS sort partitions by per-class counts that INCLUDE the dropped candidates); the same
    PostProcessor gives the exact result again on the next, fitting batch."""
    from yolo_v3_amd.utils import PostProcessor
    B, N = 3, 4000
    big, small = clustered_detections(B, N, 1500), clustered_detections(B, N, 40)
    pp = PostProcessor(B, N, 80, "cuda", max_cand=256, cap=256)
    pp._workspace(256)                                                            # allocate, then put a guard right behind it
    guard = torch.full((1 << 22,), 7, dtype=torch.uint8, device="cuda")
    out, counts = pp.run_sync_free(big.cuda(), 0.5, 0.4, False, True)
    torch.cuda.synchronize()
    host = counts.cpu()
    assert int(host[:B].max()) > 256
    with pytest.raises(_ffi.Yv3Error, match="overflow"):
        pp.to_list(out, host)
    assert int(guard.min()) == 7 and int(guard.max()) == 7
    out, counts = pp.run_sync_free(small.cuda(), 0.5, 0.4, False, True)
    got = pp.to_list(out, counts.cpu())
    want = oc.postprocess(small, 80, 0.5, 0.4, False, True)
    check_result_convention(got, want)
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_nms_properties_at_full_size():
    """Size-independent properties at config-5 scale (8 x 22743 rows, ~5k candidates per image in
    few classes): ordering, idempotence (NMS of the survivors keeps them all), batch independence."""
    B, N = 8, 22743
    u = synth.uniform01(77, 1, B * N * 8).reshape(B, N, 8)
    d = np.zeros((B, N, 85), dtype=np.float32)
    d[..., 0] = 40 + u[..., 0] * 520
    d[..., 1] = 40 + u[..., 1] * 520
    d[..., 2] = 30 + u[..., 2] * 90
    d[..., 3] = 30 + u[..., 3] * 90
    d[..., 4] = np.where(u[..., 4] < 0.25, 0.6 + 0.39 * u[..., 5], 0.1)
    cls = (u[..., 6] * 4).astype(np.int64) * 7
    bi, ri = np.meshgrid(np.arange(B), np.arange(N), indexing="ij")
    d[bi, ri, 5 + cls] = 0.85 + 0.14 * u[..., 7]
    dt = torch.from_numpy(d).cuda()
    res = postprocessing(dt, 80, 0.5, 0.4)
    assert len(res) == B
    for r in res:
        assert r.shape[0] > 100
        c, s = r[:, 6], r[:, 5]
        assert bool((c[1:] >= c[:-1]).all())                                   # classes ascending
        same = c[1:] == c[:-1]
        assert bool((s[1:][same] <= s[:-1][same]).all())                       # scores descending within a class
        iou = iou_vectorized(r[:, :4].cuda()).cpu()
        clash = (iou > 0.4) & (c[:, None] == c[None, :]) & ~torch.eye(len(r), dtype=torch.bool)
        assert not bool(clash.any())                                           # no surviving same-class overlap
    # batch independence: image order permuted -> results permuted, bit for bit
    perm = [3, 0, 7, 1, 6, 2, 5, 4]
    res_p = postprocessing(dt[perm], 80, 0.5, 0.4)
    for i, p in enumerate(perm):
        assert torch.equal(res_p[i], res[p])
    # spot-check one image against the oracle (exact)
    ref = oc.postprocess(dt[:1].cpu(), 80, 0.5, 0.4)
    assert torch.equal(res[0], ref[0])


def _grid_detections(B, N, seed, ncls=3):
    """Boxes with even integer extents on an integer grid: IOUs are ratios of small integers, so many pairs sit EXACTLY on a
    threshold like 0.5, 0.25 or float32(1/3) (strict `>`: not suppressed) or one step beside it."""
    u = synth.uniform01(seed, 3, B * N * 8).reshape(B, N, 8)
    d = np.zeros((B, N, 85), dtype=np.float32)
    d[..., 0] = 40 + np.floor(u[..., 0] * 24)
    d[..., 1] = 40 + np.floor(u[..., 1] * 24)
    d[..., 2] = 2 * (1 + np.floor(u[..., 2] * 6))
    d[..., 3] = 2 * (1 + np.floor(u[..., 3] * 6))
    d[..., 4] = 0.6 + 0.39 * u[..., 4]
    d[..., 5:] = 0.01
    cls = (u[..., 5] * ncls).astype(np.int64) * 11
    bi, ri = np.meshgrid(np.arange(B), np.arange(N), indexing="ij")
    d[bi, ri, 5 + cls] = 0.8 + 0.19 * u[..., 6]
    return d


@pytest.mark.parametrize("thr", [0.5, 0.25, float(np.float32(1.0) / np.float32(3.0)), 0.2, 0.6, 0.4])
def test_nms_pairs_exactly_on_the_threshold(thr):
    """The mask kernel decides `inter / union > thr` without dividing (exact reformulation, postproc.hip mask_kernel): on integer
    boxes thousands of pairs have an IOU equal to the threshold or a few ulp away -- every decision must equal the oracle's
    literal fp32 division + compare (reference utils.py:98-119,177-190)."""
    dt = torch.from_numpy(_grid_detections(2, 1400, 41))
    ref = oc.postprocess(dt, 80, 0.5, thr, False, True)
    res = postprocessing(dt.cuda(), 80, 0.5, thr, False, True)
    check_result_convention(res, ref)
    for r, e in zip(res, ref):
        assert torch.equal(r, e)
    assert 20 < sum(len(r) for r in res) < 2 * 1400


def test_nms_boxes_the_fast_compare_must_not_see():
    """NaN / infinite / huge coordinates and negative extents among ordinary boxes: tiles holding such a box take the literal
    IOU path; results equal the oracle bit for bit (NaN IOUs compare false, a NaN self-IOU drops the box, utils.py:182)."""
    d = _grid_detections(2, 900, 43, ncls=2)
    d[:, 5::37, 0] = np.nan
    d[:, 11::41, 2] = np.inf
    d[:, 17::43, 3] = -6.0                     # negative height: y2 < y1
    d[:, 23::47, 0] = 3e30
    d[:, 29::53, 2] = 1e25
    d[:, 31::59, 2:4] = 0.0                    # zero area
    dt = torch.from_numpy(d)
    for thr in (0.4, 0.5):
        ref = oc.postprocess(dt.clone(), 80, 0.5, thr, False, True)
        res = postprocessing(dt.cuda(), 80, 0.5, thr, False, True)
        check_result_convention(res, ref)
        for r, e in zip(res, ref):
            assert r.shape == e.shape and np.array_equal(r.numpy(), e.numpy(), equal_nan=True)
    # thresholds outside the fast compare's range (0, negative, tiny, >= 1) take the literal path everywhere
    for thr in (0.0, -0.5, 1e-42, 1.0):
        ref = oc.postprocess(dt.clone(), 80, 0.5, thr, False, True)
        res = postprocessing(dt.cuda(), 80, 0.5, thr, False, True)
        for r, e in zip(res, ref):
            assert r.shape == e.shape and np.array_equal(r.numpy(), e.numpy(), equal_nan=True)


def test_nms_long_single_class_segment():
    """One class with ~6000 candidates in one image (94 mask words, the scan's word pipeline over all its waves) next to an
    image with none and one with a handful: bit-exact vs the oracle."""
    N = 6400
    u = synth.uniform01(47, 1, 3 * N * 8).reshape(3, N, 8)
    d = np.zeros((3, N, 85), dtype=np.float32)
    d[..., 0] = 30 + u[..., 0] * 540
    d[..., 1] = 30 + u[..., 1] * 540
    d[..., 2] = 20 + u[..., 2] * 70
    d[..., 3] = 20 + u[..., 3] * 70
    d[0, :, 4] = 0.55 + 0.44 * u[0, :, 4]
    d[1, :, 4] = 0.1
    d[2, :, 4] = np.where(np.arange(N) % 500 == 3, 0.9, 0.1)
    d[..., 5 + 17] = 0.8 + 0.19 * u[..., 5]
    dt = torch.from_numpy(d)
    ref = oc.postprocess(dt, 80, 0.5, 0.4, False, True)
    res = postprocessing(dt.cuda(), 80, 0.5, 0.4, False, True)
    check_result_convention(res, ref)
    for r, e in zip(res, ref):
        assert torch.equal(r, e)
    assert len(res[0]) > 300


@pytest.mark.parametrize("C", [200, 300])
def test_postprocessing_many_classes(C):
    """Class counts beyond the 80 of COCO take other code paths: above 183 classes the filter's LDS staging is off, above 256 the
    sort runs without the per-class sub-partitions (one partition per class).  Clustered boxes, one class with ~1000
    candidates (sub-partitioned when C = 200), bit-exact vs the oracle in NMS, eval and raw mode."""
    B, N = 2, 2600
    u = synth.uniform01(700 + C, 1, B * N * 8).reshape(B, N, 8)
    d = np.zeros((B, N, 5 + C), dtype=np.float32)
    d[..., 0] = 60 + u[..., 0] * 300
    d[..., 1] = 60 + u[..., 1] * 300
    d[..., 2] = 20 + u[..., 2] * 60
    d[..., 3] = 20 + u[..., 3] * 60
    d[..., 4] = np.where(u[..., 4] < 0.6, 0.55 + 0.44 * u[..., 5], 0.2 * u[..., 5])
    d[..., 5:] = synth.uniform01(701 + C, 2, B * N * C).reshape(B, N, C) * 0.5
    cls = np.where(u[..., 6] < 0.7, C - 3, (u[..., 7] * C).astype(np.int64))
    bi, ri = np.meshgrid(np.arange(B), np.arange(N), indexing="ij")
    d[bi, ri, 5 + cls] = 0.75 + 0.24 * u[..., 7]
    dt = torch.from_numpy(d)
    for thr, is_eval, use_nms in ((0.5, False, True), (0.45, True, True), (0.5, False, False)):
        ref = oc.postprocess(dt, C, thr, 0.4, is_eval, use_nms)
        res = postprocessing(dt.cuda(), C, thr, 0.4, is_eval, use_nms)
        check_result_convention(res, ref)
        for r, e in zip(res, ref):
            assert torch.equal(r, e)
    assert max(len(r) for r in oc.postprocess(dt, C, 0.5, 0.4, False, False)) > 1000


# ----------------------------------------------------------------------------- convolutions
def _rand_cbr(cin, cout, k, s, seed):
    m = conv_bn_relu(cin, cout, k, s)
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        fan = cin * k * k
        m.conv.weight.copy_((torch.rand(m.conv.weight.shape, generator=gen) * 2 - 1) * (6.0 / fan) ** 0.5)
        m.bn.weight.copy_(torch.rand(cout, generator=gen) * 0.6 + 0.6)
        m.bn.bias.copy_(torch.rand(cout, generator=gen) * 0.4 - 0.2)
        m.bn.running_mean.copy_(torch.rand(cout, generator=gen) * 0.4 - 0.2)
        m.bn.running_var.copy_(torch.rand(cout, generator=gen) * 0.7 + 0.7)
    return m.eval()


def _ref_cbr(m, x):
    y = F.conv2d(x.double(), m.conv.weight.double(), None, m.conv.stride, m.conv.padding)
    y = F.batch_norm(y, m.bn.running_mean.double(), m.bn.running_var.double(), m.bn.weight.double(), m.bn.bias.double(), False, 0.1, 1e-5)
    return F.leaky_relu(y, 0.1)


def _run_mode(m, x_nchw, mode, residual_nchw=None):
    """conv_bn_relu `m` (already on the GPU) on NCHW fp32 input through the C-ABI in math mode `mode`."""
    sp = m._spec()
    pc = engine.pack_conv(m, sp, mode)
    B, _, H, W = x_nchw.shape
    x = engine.to_planes(x_nchw.cuda().permute(0, 2, 3, 1).contiguous(), mode)
    ho, wo = engine.out_hw(H, W, sp.k, sp.stride)
    y = engine.alloc_act(B, ho, wo, sp.cout, mode, "cuda")
    r = engine.to_planes(residual_nchw.cuda().permute(0, 2, 3, 1).contiguous(), mode) if residual_nchw is not None else None
    d = engine.make_desc(pc, x, y, B, H, W, r, dtype=mode)
    _ffi.check(_ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()))
    return engine.from_planes(y, mode).permute(0, 3, 1, 2).cpu()


CONV_CASES = [
    # cin, cout, k, stride, B, H, W    (covers every tile config, halo handling, M tails, stride 2)
    (3, 32, 3, 1, 2, 40, 56),
    (32, 64, 3, 2, 2, 40, 56),
    (64, 32, 1, 1, 3, 23, 17),
    (32, 64, 3, 1, 2, 19, 21),
    (64, 128, 3, 2, 2, 26, 26),
    (128, 64, 1, 1, 2, 13, 13),
    (128, 256, 3, 1, 5, 13, 13),
    (256, 512, 3, 2, 9, 26, 26),
    (512, 256, 1, 1, 33, 13, 13),
    (256, 128, 1, 1, 64, 26, 26),
    (1024, 512, 1, 1, 2, 13, 13),
    (512, 1024, 3, 1, 24, 13, 13),
]


@pytest.mark.parametrize("cin,cout,k,s,B,H,W", CONV_CASES)
def test_conv_bn_relu_vs_fp64(cin, cout, k, s, B, H, W):
    """darknet.py:27-44.  fp32 MFMA == fmaf chain; against an fp64 torch reference the error is
    fp32 round-off: tolerance 2e-5 * max(1,|ref|) (K up to 4608)."""
    m = _rand_cbr(cin, cout, k, s, seed=cin + cout + k)
    x = torch.rand(B, cin, H, W, generator=torch.Generator().manual_seed(1)) * 2 - 0.5
    ref = _ref_cbr(m, x)
    out = m.cuda()(x.cuda()).cpu()
    assert out.shape == ref.shape
    assert_close_rel(out, ref, 2e-5, "conv %s" % ((cin, cout, k, s),))


@pytest.mark.parametrize("cin,cout,k,s,B,H,W", [c for c in CONV_CASES if c[0] != 3] + [(64, 128, 3, 1, 70, 26, 26), (256, 256, 1, 1, 40, 26, 26)])
@pytest.mark.parametrize("mode", [_ffi.F32X3, _ffi.F32H2])
def test_conv_bf16x3_split_vs_fp64(cin, cout, k, s, B, H, W, mode):
    """Same layers in the two split modes (plane tensors, fp32 accumulate): YV3_F32_BF16X3 (exact 3-way bf16
    split, 6 MFMAs per product, dropped terms <= 2^-26) and YV3_F32_F16X2 (fp16 hi+lo, 3 MFMAs per product,
    dropped term <= 2^-22).  Both are fp32-class: the SAME round-off tolerance as the exact-fp32 kernel
    applies, 2e-5 * max(1,|ref|)."""
    m = _rand_cbr(cin, cout, k, s, seed=cin + cout + k)
    x = torch.rand(B, cin, H, W, generator=torch.Generator().manual_seed(1)) * 2 - 0.5
    ho, wo = engine.out_hw(H, W, k, s)
    res = torch.rand(B, cout, ho, wo, generator=torch.Generator().manual_seed(2)) - 0.5
    ref = _ref_cbr(m, x) + res.double()
    out = _run_mode(m.cuda(), x, mode, res)
    assert out.shape == ref.shape
    assert_close_rel(out, ref, 2e-5, "split conv mode %d %s" % (mode, (cin, cout, k, s)))


@pytest.mark.parametrize("cin,cout,k,s,B,H,W", [(512, 1024, 3, 1, 64, 13, 13), (1024, 512, 1, 1, 64, 13, 13), (256, 512, 3, 2, 50, 26, 26),
                                                (512, 1024, 3, 1, 47, 13, 13), (512, 1024, 3, 1, 4, 13, 13), (256, 512, 3, 1, 1, 26, 26),
                                                (128, 256, 3, 1, 1, 52, 52), (512, 1024, 3, 1, 1, 13, 13)])
def test_conv_stream_k_schedule(cin, cout, k, s, B, H, W):
    """Opt-in stream-K schedule (yv3_conv_desc.workspace): every CU owns an equal range of (tile, K-chunk)
    iterations; a split tile is finished by the workgroup holding its head part with the accumulators the following
    workgroups of its XCD left in the workspace -- two-way splits on launches of 1-2 rounds of tiles, many-way splits
    (a range shorter than a tile) when there are far fewer tiles than CUs; shapes outside the rule run the plain
    schedule with the workspace ignored.  Same fp32-class tolerance vs fp64 as the plain schedule,
    the hand-over flags are all cleared again, no scheduling error is flagged, and a second launch reproduces the
    first bit for bit (the split points are a function of the shape only)."""
    mode = _ffi.F32H2
    m = _rand_cbr(cin, cout, k, s, seed=cin + cout + k)
    x = torch.rand(B, cin, H, W, generator=torch.Generator().manual_seed(1)) * 2 - 0.5
    ho, wo = engine.out_hw(H, W, k, s)
    res = torch.rand(B, cout, ho, wo, generator=torch.Generator().manual_seed(2)) - 0.5
    ref = _ref_cbr(m, x) + res.double()
    mc = m.cuda()
    pc = engine.pack_conv(mc, mc._spec(), mode)
    xp = engine.to_planes(x.cuda().permute(0, 2, 3, 1).contiguous(), mode)
    rp = engine.to_planes(res.cuda().permute(0, 2, 3, 1).contiguous(), mode)
    y = engine.alloc_act(B, ho, wo, cout, mode, "cuda")
    ws = torch.zeros(_ffi.lib().yv3_conv_workspace_bytes(), dtype=torch.uint8, device="cuda")
    flags = torch.zeros(1, dtype=torch.int32, device="cuda")
    d = engine.make_desc(pc, xp, y, B, H, W, rp, dtype=mode, flags=flags, workspace=ws)
    _ffi.check(_ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()))
    out = engine.from_planes(y, mode).permute(0, 3, 1, 2).cpu()
    assert_close_rel(out, ref, 2e-5, "stream-K conv %s" % ((cin, cout, k, s, B),))
    assert int(flags.item()) == 0
    assert int(ws[-4 * 512:].view(torch.int32).abs().sum()) == 0          # every hand-over flag consumed
    first = y.clone()
    _ffi.check(_ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()))
    assert torch.equal(y, first)


@pytest.mark.parametrize("mode", [_ffi.F32, _ffi.F32X3, _ffi.F32H2])
@pytest.mark.parametrize("B,H,W", [(2, 40, 56), (1, 9, 131), (3, 64, 32), (1, 33, 260)])
def test_first_layer_all_modes_vs_fp64(B, H, W, mode):
    """yv3_conv0 (darknet.py:76): direct VALU kernel (fp32 / bf16x3 outputs) and the matrix-core kernel of the
    fp16-plane mode (K = 27 padded to 32, permuted channel rows, LDS-staged patch) on shapes with row / column
    tails and partial 128-column workgroup tiles; fp32-class tolerance 2e-5 * max(1,|ref|), and the un-split planes
    reproduce the value exactly as stored."""
    m = _rand_cbr(3, 32, 3, 1, seed=5)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3))
    ref = _ref_cbr(m, x)
    mc = m.cuda()
    pc = engine.pack_conv(mc, mc._spec(), mode)
    y = engine.alloc_act(B, H, W, 32, mode, "cuda")
    flags = torch.zeros(1, dtype=torch.int32, device="cuda")
    xg = x.cuda().contiguous()
    _ffi.check(_ffi.lib().yv3_conv0(xg.data_ptr(), pc.w.data_ptr(), pc.alpha.data_ptr(), pc.beta.data_ptr(), y.data_ptr(),
                                    B, H, W, mode, flags.data_ptr(), _ffi.stream_ptr()))
    out = engine.from_planes(y, mode).permute(0, 3, 1, 2).cpu()
    assert_close_rel(out, ref, 2e-5, "first layer mode %d" % mode)
    assert int(flags.item()) == 0
    if mode == _ffi.F32H2:                      # inputs beyond the scaled fp16 range are reported, not silently wrong
        xg[0, 1, 3, 4] = 5000.0
        _ffi.check(_ffi.lib().yv3_conv0(xg.data_ptr(), pc.w.data_ptr(), pc.alpha.data_ptr(), pc.beta.data_ptr(), y.data_ptr(),
                                        B, H, W, mode, flags.data_ptr(), _ffi.stream_ptr()))
        assert int(flags.item()) == 1


@pytest.mark.parametrize("cin,cout,k,s,B,H,W", [(32, 64, 3, 2, 2, 40, 56), (128, 256, 3, 1, 5, 13, 13), (512, 256, 1, 1, 33, 13, 13),
                                                 (256, 128, 1, 1, 64, 26, 26), (512, 1024, 3, 1, 24, 13, 13)])
def test_conv_bf16_mode_vs_fp64(cin, cout, k, s, B, H, W):
    """YV3_BF16 (BASELINE config 3: bf16 convs): bf16 tensors + weights, fp32 accumulate and epilogue.
    NOT a 1e-4 mode: operands carry 2^-9 relative rounding, so the tolerance is 2e-2 * max(1,|ref|)
    against an fp64 reference fed the same bf16-rounded input."""
    m = _rand_cbr(cin, cout, k, s, seed=cin + cout + k)
    x = (torch.rand(B, cin, H, W, generator=torch.Generator().manual_seed(1)) * 2 - 0.5).bfloat16().float()
    ref = _ref_cbr(m, x)
    out = _run_mode(m.cuda(), x, _ffi.BF16)
    assert_close_rel(out, ref, 2e-2, "bf16 conv %s" % ((cin, cout, k, s),))


@pytest.mark.parametrize("cin,cout,s,B,H,W,selected", [(256, 512, 1, 16, 38, 38, True), (128, 256, 1, 8, 76, 76, True), (256, 512, 2, 3, 37, 41, False),
                                                        (512, 1024, 1, 64, 13, 13, True), (128, 256, 1, 64, 52, 52, False)])
def test_conv_bf16_tile_variants_agree_bitwise(cin, cout, s, B, H, W, selected):
    """bf16 3x3 layers: the per-launch tile choice (conv_planes.hip: 256x128 four-wave / 256x256 / 192x256 eight-wave rolling tiles;
    the 192-row tile stages 24 pixel rows per wave = one DMA piece and a half) only changes the schedule -- same K order, so every
    forced tile must equal the library's own choice BIT FOR BIT, M tails included (23104 = 120 x 192 + 64 rows; 3 x 19 x 21 rows);
    `selected`: shapes whose tile counts make the shipped rule take the 192-row tile (switching the rule off through tune[1] bit 4
    must change nothing either).  One small shape is also checked against fp64 (same tolerance as test_conv_bf16_mode_vs_fp64)."""
    m = _rand_cbr(cin, cout, 3, s, seed=cin + cout + 3).cuda()
    sp = m._spec()
    pc = engine.pack_conv(m, sp, _ffi.BF16)
    xn = (torch.rand(B, cin, H, W, generator=torch.Generator().manual_seed(5)) * 2 - 0.5).bfloat16().float()
    x = engine.to_planes(xn.cuda().permute(0, 2, 3, 1).contiguous(), _ffi.BF16)
    ho, wo = engine.out_hw(H, W, 3, s)
    outs = {}
    for code in (0, 7, 8, 11, 13, 14, 15, 16, "rule off", "rolling loop", "3-deep ring"):
        y = engine.alloc_act(B, ho, wo, cout, _ffi.BF16, "cuda")
        y.zero_()
        d = engine.make_desc(pc, x, y, B, H, W, None, dtype=_ffi.BF16)
        if code == "rule off":
            d.tune[1] = 16
        elif code == "rolling loop":                    # round 5: the shipped 192x256 / 256x256 tiles run the ping-pong loop on a 4-deep ring
            d.tune[1] = 512                             # (bit 9: the round-4 rolling loop; bit 10: ping-pong on the 3-deep ring)
        elif code == "3-deep ring":
            d.tune[1] = 1024
        else:
            d.options = (d.options & ~(0xff << 8)) | (code << 8)      # (13, 14 / 15, 16: the 256- / 192-row tile with the ping-pong loop, 3- / 4-deep ring)
        _ffi.check(_ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()))
        outs[code] = y
    torch.cuda.synchronize()
    for code, y in outs.items():
        assert torch.equal(y, outs[0]), "tile %s differs from the shipped selection" % (code,)
    if B * ho * wo <= 4096:
        out = engine.from_planes(outs[11], _ffi.BF16).permute(0, 3, 1, 2).cpu()
        assert_close_rel(out, _ref_cbr(m.cpu(), xn), 2e-2, "bf16 192-row tile %s" % ((cin, cout, s),))


@pytest.mark.parametrize("cin,cout,k,s,B,H,W,res", [(128, 256, 3, 1, 64, 52, 52, True), (64, 128, 3, 1, 7, 104, 104, True), (256, 512, 3, 2, 9, 37, 41, False),
                                                      (512, 256, 1, 1, 33, 26, 26, False), (256, 128, 1, 1, 16, 52, 52, False), (512, 1024, 3, 1, 32, 13, 13, True),
                                                      (128, 256, 3, 1, 1, 11, 17, True)])
def test_conv_w4_tile_equals_eight_wave_tile_bitwise(cin, cout, k, s, B, H, W, res):
    """Round 5: the four-wave 192x128 tile with two workgroups per CU (csrc/conv_planes_w4.hip; tile code 12, and the shipped rule from
    three quarters of a workgroup per CU upwards) only changes the schedule: same K order and the same three-product order per
    k-step as the eight-wave 256x128 / 128x128 tiles (codes 1, 2), so the outputs must agree BIT FOR BIT -- 3x3 stride 1 / 2, 1x1,
    with and without residual, M tails (173056 = 901 x 192 + 64 rows; 9 x 19 x 21 rows; a single 11 x 17 image: one partial tile) --
    and the default selection must equal them too; one small shape is also checked against fp64."""
    m = _rand_cbr(cin, cout, k, s, seed=cin + cout + k).cuda()
    sp = m._spec()
    pc = engine.pack_conv(m, sp, _ffi.F32H2)
    xn = torch.rand(B, cin, H, W, generator=torch.Generator().manual_seed(5)) * 2 - 0.5
    x = engine.to_planes(xn.cuda().permute(0, 2, 3, 1).contiguous(), _ffi.F32H2)
    ho, wo = engine.out_hw(H, W, k, s)
    rn = (torch.rand(B, cout, ho, wo, generator=torch.Generator().manual_seed(6)) - 0.5) if res else None
    r = engine.to_planes(rn.cuda().permute(0, 2, 3, 1).contiguous(), _ffi.F32H2) if res else None
    outs = {}
    for code in (0, 12, 1, 2, "rule off"):
        y = engine.alloc_act(B, ho, wo, cout, _ffi.F32H2, "cuda")
        y.zero_()
        d = engine.make_desc(pc, x, y, B, H, W, r, dtype=_ffi.F32H2)
        if code == "rule off":
            d.tune[1] = 32
        else:
            d.options = (d.options & ~(0xff << 8)) | (code << 8)
        _ffi.check(_ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()))
        outs[code] = y
    torch.cuda.synchronize()
    for code, y in outs.items():
        assert torch.equal(y, outs[12]), "tile %s differs from the four-wave 192x128 tile" % (code,)
    if B * ho * wo <= 8192:
        out = engine.from_planes(outs[12], _ffi.F32H2).permute(0, 3, 1, 2).cpu()
        ref = _ref_cbr(m.cpu(), xn) + (rn.double() if res else 0.0)
        assert_close_rel(out, ref, 2e-5, "w4 tile %s" % ((cin, cout, k, s),))


@pytest.mark.parametrize("cin,cout,B,H,W", [(128, 256, 5, 13, 13), (64, 128, 70, 26, 26), (32, 64, 2, 19, 21), (512, 1024, 24, 13, 13), (256, 512, 33, 26, 26)])
def test_conv_k3s1_tap_reuse_kernel(cin, cout, B, H, W, monkeypatch):
    """Opt-in 3x3/stride-1 kernel that stages ONE activation tile per (kh, channel chunk) and reuses it for
    the three kw taps (csrc/conv_planes_k3s1.hip): must equal the generic plane kernel to fp32 round-off
    (same products, different accumulation order) and the fp64 reference at 2e-5."""
    m = _rand_cbr(cin, cout, 3, 1, seed=cin + cout)
    x = torch.rand(B, cin, H, W, generator=torch.Generator().manual_seed(1)) * 2 - 0.5
    res = torch.rand(B, cout, H, W, generator=torch.Generator().manual_seed(2)) - 0.5
    ref = _ref_cbr(m, x) + res.double()
    m = m.cuda()
    generic = _run_mode(m, x, _ffi.F32X3, res)
    monkeypatch.setenv("YV3_MEASURE", "1")                  # tuning overrides are honoured in measurement sessions only
    monkeypatch.setenv("YV3_K3S1", "1")
    reuse = _run_mode(m, x, _ffi.F32X3, res)
    assert_close_rel(reuse, ref, 2e-5, "k3s1 conv")
    assert_close_rel(reuse, generic, 2e-5, "k3s1 vs generic")   # different K order -> fp32 round-off only


def test_plane_split_is_exact():
    """fp32 -> 3 bf16 planes -> fp32 is the identity (8+8+8 mantissa bits), including tiny/huge values."""
    g = torch.Generator().manual_seed(5)
    x = torch.cat((torch.randn(100000, generator=g), torch.randn(1000, generator=g) * 1e-30,
                   torch.randn(1000, generator=g) * 1e30, torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38]))).cuda()
    back = engine.from_planes(engine.to_planes(x.view(1, 1, 1, -1), _ffi.F32X3), _ffi.F32X3).view(-1)
    assert torch.equal(back, x)


def test_residual_block_and_upsample_concat():
    """darknet.py:46-53 (residual add in the epilogue) and :153-162 (upsample+cat fused in the gather)."""
    blk = res_layer(64)
    blk.conv1, blk.conv2 = _rand_cbr(64, 32, 1, 1, 5), _rand_cbr(32, 64, 3, 1, 6)
    x = torch.rand(3, 64, 21, 19) - 0.3
    ref = x.double() + _ref_cbr(blk.conv2, _ref_cbr(blk.conv1, x).float())
    out = blk.cuda()(x.cuda()).cpu()
    assert_close_rel(out, ref, 2e-5, "res_layer")
    # fused dual-source 1x1: conv(cat(up2x(a), b)) with a [B,64,h,w], b [B,128,2h,2w]
    cons = _rand_cbr(192, 96, 1, 1, 7)
    a, b = torch.rand(2, 64, 7, 9) - 0.5, torch.rand(2, 128, 14, 18) - 0.5
    cat = torch.cat((F.interpolate(a, scale_factor=2, mode="nearest"), b), 1)
    ref = _ref_cbr(cons, cat)
    pc = engine.pack_conv(cons.cuda(), cons._spec(), _ffi.F32)
    a_n, b_n = a.cuda().permute(0, 2, 3, 1).contiguous(), b.cuda().permute(0, 2, 3, 1).contiguous()
    y = torch.empty(2, 14, 18, 96, device="cuda")
    d = engine.make_desc(pc, a_n, y, 2, 14, 18, x2=b_n, cin_up=64)
    _ffi.check(_ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()))
    assert_close_rel(y.permute(0, 3, 1, 2).cpu(), ref, 2e-5, "upsample+concat conv")
    # stand-alone UpsampleGroup keeps the reference's channel order: upsampled first
    up = UpsampleGroup(64)
    up.conv = _rand_cbr(64, 32, 1, 1, 8)
    t = torch.rand(2, 32, 14, 18)
    out = up.cuda()(a.cuda(), t.cuda()).cpu()
    ref = torch.cat((F.interpolate(_ref_cbr(up.conv.cpu(), a).float(), scale_factor=2, mode="nearest"), t), 1)
    assert_close_rel(out, ref, 2e-5, "UpsampleGroup")
    # ... and its tail (nearest x2 + concat) is ONE HIP launch, bit-exact data movement (round 5; darknet.py:159-162), odd sizes included
    for (Bq, cu, ct, hq, wq) in ((2, 32, 32, 7, 9), (1, 3, 0, 5, 5), (3, 16, 40, 13, 13)):
        u, tl = torch.rand(Bq, cu, hq, wq).cuda(), torch.rand(Bq, ct, 2 * hq, 2 * wq).cuda()
        o = torch.full((Bq, cu + ct, 2 * hq, 2 * wq), float("nan"), device="cuda")
        _ffi.check(_ffi.lib().yv3_upsample2x_concat(u.data_ptr(), tl.data_ptr(), o.data_ptr(), Bq, cu, ct, hq, wq, _ffi.stream_ptr()))
        want = torch.cat((F.interpolate(u.cpu(), scale_factor=2, mode="nearest"), tl.cpu()), 1)
        assert torch.equal(o.cpu(), want)


def test_head_conv_255_and_asymmetric_layout():
    """Plain head conv (darknet.py:118: bias, no BN/activation, cout=255 padded to 256) on inputs
    where every (pixel, channel) is distinct: catches row/col swaps in the MFMA C layout."""
    head = torch.nn.Conv2d(64, 255, 1)
    with torch.no_grad():
        head.weight.copy_(torch.arange(255 * 64).float().view(255, 64, 1, 1) % 17 - 8.0)
        head.bias.copy_(torch.arange(255).float() / 8)
    x = (torch.arange(2 * 64 * 5 * 7).float().view(2, 64, 5, 7) % 13) - 6.0
    ref = F.conv2d(x.double(), head.weight.double(), head.bias.double())
    grp = PreDetectionConvGroup(64, 32, num_conv=0, numClass=80)
    grp.mlist = torch.nn.ModuleList([head])
    out = grp.cuda()(x.cuda()).cpu()
    assert out.shape == (2, 255, 5, 7)
    assert torch.equal(out.double(), ref)            # small integers: exact in fp32


# ----------------------------------------------------------------------------- neighbours of the path
def test_correct_yolo_boxes_bit_exact_vs_reference(golden_dir):
    """boundingbox.py:95-149 on the GPU: un-letterbox / un-resize + clip + xywh, bitwise equal to the reference."""
    from yolo_v3_amd import correct_yolo_boxes, letterbox_reverse, rescale_bbox
    g = np.load(os.path.join(golden_dir, "neighbours.npz"))
    boxes = torch.from_numpy(g["boxes"])
    for ci, (ow, oh, iw, ih) in enumerate(g["cases"].tolist()):
        for lb in (0, 1):
            out = correct_yolo_boxes(boxes.cuda(), ow, oh, iw, ih, bool(lb))
            assert out.is_cuda and np.array_equal(out.cpu().numpy(), g["out_%d_%d" % (ci, lb)]), (ci, lb)
        assert np.array_equal(letterbox_reverse(boxes, ow, oh, iw, ih).numpy(), g["xyxy_%d" % ci])       # CPU in -> CPU out
        assert np.array_equal(rescale_bbox(boxes.cuda(), ow, oh, iw, ih).cpu().numpy(), g["rescale_%d" % ci])
    assert len(correct_yolo_boxes(torch.zeros(0, 4), 10, 10, 416, 416, True)) == 0


def test_letterbox_vs_oracle(golden_dir):
    """utils.py:34-72 on the GPU: cv2.resize(INTER_CUBIC) as OpenCV's fixed-point 8-bit path defines it (11-bit
    short coefficients, int32 horizontal pass, (sum + 2^21) >> 22 vertical pass), restated in oracle_cpu from
    OpenCV's resize.cpp.  PARITY UNPINNED (cv2 absent: no golden vector can be produced here); the kernel equals the
    oracle's restatement BIT FOR BIT on up- and down-scales, including clipping overshoot; geometry exact."""
    from yolo_v3_amd import letterbox_batch, letterbox_transforms
    imgs = [(synth.uniform01(31 + i, 9, h * w * 3).reshape(h, w, 3) * 255).astype(np.uint8) for i, (h, w) in
            enumerate([(452, 602), (300, 200), (100, 640), (416, 416), (37, 53), (1080, 1920)])]
    # smooth structure so that bicubic overshoot / clipping paths are exercised too
    imgs[0][100:200, 150:400] = 255; imgs[0][250:300, :] = 0
    batch, trans = letterbox_batch(imgs, (416, 416))
    assert batch.shape == (6, 3, 416, 416) and batch.is_cuda
    for i, im in enumerate(imgs):
        ref = oc.letterbox_image(im, (416, 416))
        got = batch[i].cpu()
        assert torch.equal(got, ref), "image %d: %d pixels differ" % (i, int((got != ref).sum()))
        bw, bh, bx, by, ratio = letterbox_transforms((im.shape[1], im.shape[0]), (416, 416))
        assert trans[i].tolist()[:4] == [bw, bh, bx, by]
        pad = torch.ones(416, 416, dtype=torch.bool); pad[by:by + bh, bx:bx + bw] = False
        assert torch.equal(got[:, pad], torch.full_like(got[:, pad], 128.0 / 255.0))
    b608, _ = letterbox_batch(imgs[:2], (608, 608))
    for i in range(2):
        assert torch.equal(b608[i].cpu(), oc.letterbox_image(imgs[i], (608, 608)))
    # 602x452 -> 416x312 at y offset 52 (SURVEY config 1)
    assert trans[0].tolist()[:4] == [416, 312, 0, 52]


def test_plain_resize_vs_oracle():
    """load_image(mode='resize') (utils.py:68-71: cv2.resize(img, dim), INTER_LINEAR) on the GPU, bit for bit against
    the oracle's restatement of OpenCV's fixed-point bilinear path (PARITY UNPINNED: cv2 absent), incl. the exact-2x
    shrink that cv::resize reroutes to INTER_AREA and an up-scale."""
    from yolo_v3_amd import resize_batch, load_image
    shapes = [(452, 602), (832, 832), (100, 640), (416, 416), (37, 53), (1080, 1920)]
    imgs = [(synth.uniform01(41 + i, 9, h * w * 3).reshape(h, w, 3) * 255).astype(np.uint8) for i, (h, w) in enumerate(shapes)]
    batch = resize_batch(imgs, (416, 416))
    assert batch.shape == (6, 3, 416, 416) and batch.is_cuda
    for i, im in enumerate(imgs):
        ref = oc.resize_image(im, (416, 416))
        assert torch.equal(batch[i].cpu(), ref), "image %d: %d values differ" % (i, int((batch[i].cpu() != ref).sum()))
    one, tr = load_image(imgs[0], mode="resize", dim=(608, 352))
    assert tr is None and torch.equal(one.cpu(), oc.resize_image(imgs[0], (608, 352)))



@pytest.mark.parametrize("mode", [_ffi.F32H2, _ffi.BF16])
@pytest.mark.parametrize("B,H,W", [(2, 416, 416), (1, 608, 608), (3, 320, 480), (2, 96, 64), (5, 32, 32)])
def test_fused_front_equals_two_launches_bitwise(B, H, W, mode):
    """csrc/conv_front.hip (feature.mlist.0 + feature.mlist.1 in one launch, the first layer's activation kept in LDS) writes
    BIT FOR BIT what yv3_conv0 followed by yv3_conv2d writes (same products, same order), on square / non-square / tiny
    inputs incl. all four image borders; whole-net detections are therefore identical too; saturation is still reported."""
    from yolo_v3_amd import YoloNet, WeightManager
    stream = synth.weight_stream()
    net = YoloNet((W, H)).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    net = net.cuda()
    x = torch.from_numpy(synth.images(B, max(H, W), 7)[:, :, :H, :W].copy()).cuda()
    net.math_mode = mode
    eng = net.engine(mode)
    outs, dets = [], []
    for fused in (False, True):
        eng.fuse_front, eng._plans = fused, {}
        try:
            d, plan = eng.forward(x)
            assert plan.fused_front == fused
            outs.append(plan.layer_out["feature.mlist.1"].clone())
            dets.append(d.clone())
        finally:
            eng.fuse_front, eng._plans = True, {}
    assert outs[0].shape == (2 if mode == _ffi.F32H2 else 1, B, H // 2, W // 2, 64)
    assert torch.equal(outs[0], outs[1]), "%d elements differ" % int((outs[0] != outs[1]).sum())
    assert torch.equal(dets[0], dets[1])
    net.strict_range = True                                     # (default: fall back to F32X3, tests/test_gpu_e2e.py)
    with pytest.raises(_ffi.Yv3Error, match="fp16 range|fp16 matrix cores"):
        net.forward_cat(x * 1e4)                               # |x| * 16 leaves the fp16 range of the first layer's operands



@pytest.mark.parametrize("mode", [_ffi.F32H2, _ffi.BF16])
@pytest.mark.parametrize("B,H,W", [(2, 416, 416), (1, 608, 608), (3, 320, 480), (2, 96, 64), (5, 32, 32)])
def test_fused_res64_equals_two_launches_bitwise(B, H, W, mode):
    """csrc/conv_res64.hip (feature.mlist.2 = 1x1 64->32 + 3x3 32->64 + residual add in one launch, the 32-channel map kept
    in LDS) writes BIT FOR BIT what the two yv3_conv2d launches write, incl. all image borders; whole-net detections equal."""
    from yolo_v3_amd import YoloNet, WeightManager
    stream = synth.weight_stream()
    net = YoloNet((W, H)).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    net = net.cuda()
    x = torch.from_numpy(synth.images(B, max(H, W), 8)[:, :, :H, :W].copy()).cuda()
    net.math_mode = mode
    eng = net.engine(mode)
    outs, dets = [], []
    for fused in (False, True):
        eng.fuse_res64, eng._plans = fused, {}
        try:
            d, plan = eng.forward(x)
            assert plan.fused_res64 == fused and plan.first_desc == (3 if fused else 1)
            outs.append(plan.layer_out["feature.mlist.2.conv2"].clone())
            dets.append(d.clone())
        finally:
            eng.fuse_res64, eng._plans = True, {}
    assert outs[0].shape == (2 if mode == _ffi.F32H2 else 1, B, H // 2, W // 2, 64)
    assert torch.equal(outs[0], outs[1]), "%d elements differ" % int((outs[0] != outs[1]).sum())
    assert torch.equal(dets[0], dets[1])


@pytest.mark.parametrize("B,H,W", [(2, 416, 416), (1, 608, 608), (3, 320, 480), (2, 96, 64), (5, 32, 32)])
def test_fused_front_f32_equals_two_launches_bitwise(B, H, W):
    """csrc/conv_front_f32.hip, the exact-fp32 twin of the fused front: feature.mlist.0 + feature.mlist.1 in one launch write BIT FOR BIT
    what yv3_conv0 (vector-ALU fma chain) followed by yv3_conv2d write in YV3_F32, incl. all four image borders; detections equal."""
    from yolo_v3_amd import YoloNet, WeightManager
    stream = synth.weight_stream()
    net = YoloNet((W, H)).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    net = net.cuda()
    x = torch.from_numpy(synth.images(B, max(H, W), 10)[:, :, :H, :W].copy()).cuda()
    net.math_mode = _ffi.F32
    eng = net.engine(_ffi.F32)
    outs, dets = [], []
    for fused in (False, True):
        eng.fuse_front, eng._plans = fused, {}
        try:
            d, plan = eng.forward(x)
            assert plan.fused_front == fused and plan.first_desc == (3 if fused else 0)
            if fused:
                plan.layer_out["feature.mlist.1"].fill_(float("nan"))          # every element must be written
                d, plan = eng.forward(x)
            outs.append(plan.layer_out["feature.mlist.1"].clone())
            dets.append(d.clone())
        finally:
            eng.fuse_front, eng._plans = True, {}
    assert outs[0].shape == (B, H // 2, W // 2, 64) and outs[0].dtype == torch.float32
    assert torch.isfinite(outs[1]).all()
    assert torch.equal(outs[0], outs[1]), "%d elements differ, max |d| %g" % (int((outs[0] != outs[1]).sum()), float((outs[0] - outs[1]).abs().max()))
    assert torch.equal(dets[0], dets[1])


@pytest.mark.parametrize("B,H,W", [(2, 416, 416), (1, 608, 608), (3, 320, 480), (2, 96, 64), (5, 32, 32)])
def test_fused_res64_f32_equals_two_launches_bitwise(B, H, W):
    """csrc/conv_res64_f32.hip, the exact-fp32 twin: feature.mlist.2 in one launch writes BIT FOR BIT what the two yv3_conv2d
    launches of YV3_F32 write (same products, same k pairing, same order), incl. all image borders; detections equal."""
    from yolo_v3_amd import YoloNet, WeightManager
    stream = synth.weight_stream()
    net = YoloNet((W, H)).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    net = net.cuda()
    x = torch.from_numpy(synth.images(B, max(H, W), 9)[:, :, :H, :W].copy()).cuda()
    net.math_mode = _ffi.F32
    eng = net.engine(_ffi.F32)
    outs, mids, dets = [], [], []
    for fused in (False, True):
        eng.fuse_res64, eng._plans = fused, {}
        try:
            d, plan = eng.forward(x)
            assert plan.fused_res64 == fused and plan.fused_front and plan.first_desc == (3 if fused else 1)
            if fused:
                plan.layer_out["feature.mlist.2.conv2"].fill_(float("nan"))          # every element must be written
                d, plan = eng.forward(x)
            outs.append(plan.layer_out["feature.mlist.2.conv2"].clone())
            dets.append(d.clone())
        finally:
            eng.fuse_res64, eng._plans = True, {}
    assert outs[0].shape == (B, H // 2, W // 2, 64) and outs[0].dtype == torch.float32
    assert torch.isfinite(outs[1]).all()
    assert torch.equal(outs[0], outs[1]), "%d elements differ, max |d| %g" % (int((outs[0] != outs[1]).sum()), float((outs[0] - outs[1]).abs().max()))
    assert torch.equal(dets[0], dets[1])


def _run_wino(m, x_nchw, residual_nchw=None, even=False):
    """conv_bn_relu `m` (3x3, stride 1, on the GPU) through the Winograd F(2x2,3x3) form of the fp16-plane kernel."""
    mode = _ffi.F32H2
    sp = m._spec()
    pc = engine.pack_conv(m, sp, mode, winograd=True)
    assert pc.w_wino is not None
    B, _, H, W = x_nchw.shape
    x = engine.to_planes(x_nchw.cuda().permute(0, 2, 3, 1).contiguous(), mode)
    y = engine.alloc_act(B, H, W, sp.cout, mode, "cuda")
    y.fill_(float("nan"))
    r = engine.to_planes(residual_nchw.cuda().permute(0, 2, 3, 1).contiguous(), mode) if residual_nchw is not None else None
    ws = torch.zeros(_ffi.lib().yv3_wino_workspace_bytes(B, H, W, sp.cin), dtype=torch.uint8, device="cuda")
    flags = torch.zeros(1, dtype=torch.int32, device="cuda")
    d = engine.make_desc(pc, x, y, B, H, W, r, dtype=mode, wino_ws=ws, flags=flags)
    assert d.w_wino
    d.options |= _ffi.OPT_WINO_ALWAYS | (_ffi.OPT_WINO_EVEN if even else 0)
    for _ in range(2):                                   # twice: the hand-over flags of the even schedule must be left reset
        y.fill_(float("nan"))
        _ffi.check(_ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()))
    assert int(flags.item()) == 0, "kernel status word %d" % int(flags.item())
    return engine.from_planes(y, mode).permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("cin,cout,B,H,W,res", [(256, 512, 3, 26, 26, True), (512, 1024, 5, 13, 13, True), (256, 512, 2, 38, 38, False),
                                                (512, 1024, 2, 19, 19, True), (256, 128, 1, 6, 10, False), (256, 256, 2, 7, 5, True),
                                                (256, 512, 24, 26, 26, True), (512, 1024, 40, 13, 13, True)])
def test_winograd_conv_vs_fp64(cin, cout, B, H, W, res):
    """Winograd F(2x2,3x3) form of conv_bn_relu(3x3, s1) (+ residual) in the fp16 hi+lo plane mode: even and ODD pictures
    (13x13, 19x19, 7x5: the last tile row / column hangs over the edge), M tails, every output written exactly once.
    Against fp64: 2e-5 * max(1,|ref|) (the direct kernels' bar; measured ~2x the direct scheme's error,
    tools/winograd_numerics.py), and within 2e-5 of the direct fp16-plane kernel."""
    m = _rand_cbr(cin, cout, 3, 1, seed=cin + cout + H)
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.rand(B, cin, H, W, generator=g) * 2 - 0.5
    r = (torch.rand(B, cout, H, W, generator=g) - 0.5) if res else None
    ref = _ref_cbr(m, x) + (r.double() if res else 0)
    mc = m.cuda()
    out = _run_wino(mc, x, r)
    assert torch.isfinite(out).all(), "an output element was not written"
    e = assert_close_rel(out, ref, 2e-5, "winograd conv %s" % ((cin, cout, H, W),))
    direct = _run_mode(mc, x, _ffi.F32H2, r)
    e_d = float(rel_err(direct, ref).max())
    # the opt-in even schedule (YV3_OPT_WINO_EVEN: tiles split between positions, partial outputs handed over in L2): same
    # products, another summation order
    ev = _run_wino(mc, x, r, even=True)
    assert torch.isfinite(ev).all()
    e_t = assert_close_rel(ev, ref, 2e-5, "winograd (even schedule) %s" % ((cin, cout, H, W),))
    print("winograd %s: err vs fp64 %.3g tile schedule, %.3g even schedule (direct kernel %.3g)" % ((cin, cout, B, H, W), e, e_t, e_d))
    assert_close_rel(out, direct, 2e-5, "winograd vs direct")
    assert_close_rel(out, ev, 1e-5, "even vs tile schedule")


@pytest.mark.parametrize("cin,cout,B,H,W,res", [(64, 128, 2, 20, 28, True), (128, 256, 2, 13, 13, True), (256, 512, 3, 26, 26, False),
                                                (512, 1024, 2, 13, 13, True), (256, 128, 1, 7, 5, True)])
def test_winograd_f32_conv_vs_fp64(cin, cout, B, H, W, res):
    """fp32-MFMA mode (YV3_F32): the Winograd F(2x2,3x3) form of conv_bn_relu(3x3, s1) (+ residual) -- fp32 transforms, fp32
    MFMA, fp32 fold -- against fp64 (2e-5 * max(1,|ref|), the direct kernel's bar) and against the direct fp32 kernel."""
    mode = _ffi.F32
    m = _rand_cbr(cin, cout, 3, 1, seed=cin + cout + H)
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.rand(B, cin, H, W, generator=g) * 2 - 0.5
    r = (torch.rand(B, cout, H, W, generator=g) - 0.5) if res else None
    ref = _ref_cbr(m, x) + (r.double() if res else 0)
    mc = m.cuda()
    sp = mc._spec()
    pc = engine.pack_conv(mc, sp, mode, winograd=True, winograd4=False)
    assert pc.w_wino is not None and pc.w_wino4 is None
    xg = x.cuda().permute(0, 2, 3, 1).contiguous()
    rg = r.cuda().permute(0, 2, 3, 1).contiguous() if res else None
    y = torch.full((B, H, W, cout), float("nan"), device="cuda")
    ws = torch.zeros(_ffi.lib().yv3_wino_workspace_bytes(B, H, W, cin), dtype=torch.uint8, device="cuda")
    d = engine.make_desc(pc, xg, y, B, H, W, rg, dtype=mode, wino_ws=ws)
    d.options |= _ffi.OPT_WINO_ALWAYS
    assert d.w_wino
    _ffi.check(_ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()))
    out = y.permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(out).all(), "an output element was not written"
    e = assert_close_rel(out, ref, 2e-5, "fp32 winograd conv %s" % ((cin, cout, H, W),))
    direct = _run_mode(mc, x, mode, r)
    print("fp32 winograd %s: err vs fp64 %.3g (direct kernel %.3g)" % ((cin, cout, B, H, W), e, float(rel_err(direct, ref).max())))
    assert_close_rel(out, direct, 2e-5, "fp32 winograd vs direct")
    # round 5: the stage's two tiles -- eight waves 128x128 (tune[0] = 9) and four waves 64x128, two workgroups per CU (tune[0] = 8) -- keep
    # the same K order per accumulator: bit-identical to each other and to the shipped per-launch choice
    outs = []
    for code in (9, 8):
        y2 = torch.full((B, H, W, cout), float("nan"), device="cuda")
        d2 = engine.make_desc(pc, xg, y2, B, H, W, rg, dtype=mode, wino_ws=ws)
        d2.options |= _ffi.OPT_WINO_ALWAYS
        d2.tune[0] = code
        _ffi.check(_ffi.lib().yv3_conv2d(d2, _ffi.stream_ptr()))
        outs.append(y2)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], y), "the fp32 Winograd stage's tiles differ"


@pytest.mark.parametrize("cin,cout,B,H,W,res", [(64, 128, 2, 20, 28, True), (128, 256, 2, 13, 13, True), (256, 512, 3, 26, 26, False),
                                                (512, 1024, 2, 13, 13, True), (256, 128, 1, 7, 5, True), (128, 256, 3, 52, 52, True),
                                                (64, 64, 1, 4, 4, False), (128, 192, 5, 19, 19, True), (256, 512, 40, 13, 13, True),
                                                (128, 256, 1, 1, 9, True), (384, 128, 2, 6, 3, False)])
def test_winograd4_f32_conv_vs_fp64(cin, cout, B, H, W, res):
    """fp32-MFMA mode (YV3_F32): the Winograd F(4x4,3x3) form (csrc/conv_wino4_f32.hip; points 0, 1, -1, 1/2, -2, inf) of
    conv_bn_relu(3x3, s1) (+ residual): pictures whose sides are and are not multiples of 4 (13, 19, 26, 7x5, 1x9: the last tile row / column
    hangs over the edge by 1-3 pixels), tile counts that are not multiples of the workgroup's 32 (M tails), 64-wide channel blocks, 12 chunks per
    position (cin 384), every output written exactly once (NaN-filled buffer).  Against fp64: 3e-5 * max(1,|ref|) (direct kernel: 2e-5; the form's
    transforms add ~1.5x, tools/winograd_f32_gate.py), against the direct fp32 kernel and the F(2x2) form: 3e-5.  The form query must say 2."""
    mode = _ffi.F32
    m = _rand_cbr(cin, cout, 3, 1, seed=cin + cout + H)
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.rand(B, cin, H, W, generator=g) * 2 - 0.5
    r = (torch.rand(B, cout, H, W, generator=g) - 0.5) if res else None
    ref = _ref_cbr(m, x) + (r.double() if res else 0)
    mc = m.cuda()
    sp = mc._spec()
    pc = engine.pack_conv(mc, sp, mode, winograd=True) if cout % 128 == 0 else None
    if pc is None:                       # (engine.wino_eligible asks for cout % 128 -- the F(2x2) stage's tile; the F(4x4) image needs cout % 64)
        pc = engine.pack_conv(mc, sp, mode)
        pc.w_wino4 = engine.pack_wino4(mc.conv.weight.detach().float().contiguous(), sp)
    assert pc.w_wino4 is not None
    xg = x.cuda().permute(0, 2, 3, 1).contiguous()
    rg = r.cuda().permute(0, 2, 3, 1).contiguous() if res else None
    y = torch.full((B, H, W, cout), float("nan"), device="cuda")
    ws = torch.zeros(max(_ffi.lib().yv3_wino_workspace_bytes(B, H, W, cin), _ffi.lib().yv3_wino4_workspace_bytes(B, H, W, cin)), dtype=torch.uint8, device="cuda")
    d = engine.make_desc(pc, xg, y, B, H, W, rg, dtype=mode, wino_ws=ws)
    if not d.w_wino4:                    # (make_desc binds the Winograd pointers for layers that carry w_wino)
        d.w_wino4 = pc.w_wino4.data_ptr(); d.wino_ws = ws.data_ptr(); d.wino_ws_bytes = ws.numel()
    d.options |= _ffi.OPT_WINO_ALWAYS
    assert _ffi.lib().yv3_conv2d_form(d) == 2
    _ffi.check(_ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()))
    out = y.permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(out).all(), "an output element was not written"
    e = assert_close_rel(out, ref, 3e-5, "fp32 winograd F(4x4) conv %s" % ((cin, cout, H, W),))
    direct = _run_mode(mc, x, mode, r)
    print("fp32 winograd F(4x4) %s: err vs fp64 %.3g (direct kernel %.3g)" % ((cin, cout, B, H, W), e, float(rel_err(direct, ref).max())))
    assert_close_rel(out, direct, 3e-5, "fp32 winograd F(4x4) vs direct")
    # too small a workspace is an error, not a crash; tune[0] == 10 opts the descriptor out of the form
    d.wino_ws_bytes = _ffi.lib().yv3_wino4_workspace_bytes(B, H, W, cin) - 4
    assert _ffi.lib().yv3_conv2d(d, _ffi.stream_ptr()) == -3
    d.wino_ws_bytes = ws.numel()
    d.tune[0] = 10
    assert _ffi.lib().yv3_conv2d_form(d) != 2


@pytest.mark.parametrize("cin,cout,B,H,W", [(512, 1024, 1, 13, 13), (256, 512, 4, 26, 26), (128, 256, 2, 52, 52), (64, 128, 13, 104, 104),
                                            (256, 128, 3, 19, 19)])
def test_winograd4_f32_even_schedule(cin, cout, B, H, W):
    """The F(4x4,3x3) stage's even schedule (round 6; include/yv3.h, yv3_conv_desc.w_wino4): tail items cut into 2 / 3 / 6 ranges of patch rows,
    one workgroup each, summed through the hand-over area -- against one item per workgroup (tune[1] = 1).  Six parts (one row each) add the same
    products in the same order: bit-identical; two / three parts re-associate the row sums: within 2e-6 * max|y|.  The status word stays 0, every
    hand-over flag is back to zero, a second launch gives the same bits, YV3_OPT_WINO4_TILES means one item per workgroup, and -- 64 -> 128 at
    104 x 104, 13 images: 550 items -- a tail behind a full round of the chip (512 whole-item workgroups + 38 items cut)."""
    mode = _ffi.F32
    lib = _ffi.lib()
    m = _rand_cbr(cin, cout, 3, 1, seed=cin + cout + H).cuda()
    sp = m._spec()
    pc = engine.pack_conv(m, sp, mode)
    pc.w_wino4 = engine.pack_wino4(m.conv.weight.detach().float().contiguous(), sp)
    g = torch.Generator().manual_seed(H * 31 + W + B)
    xg = (torch.rand(B, H, W, cin, generator=g) * 2 - 0.5).cuda()
    rg = (torch.rand(B, H, W, cout, generator=g) - 0.5).cuda()
    ws = torch.zeros(lib.yv3_wino4_workspace_bytes(B, H, W, cin), dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")

    def run(tune1, tune2, options=0):
        y = torch.full((B, H, W, cout), float("nan"), device="cuda")
        d = engine.make_desc(pc, xg, y, B, H, W, rg, dtype=mode, flags=status)
        d.w_wino4 = pc.w_wino4.data_ptr(); d.wino_ws = ws.data_ptr(); d.wino_ws_bytes = ws.numel()
        d.options |= _ffi.OPT_WINO_ALWAYS | options
        d.tune[1], d.tune[2] = tune1, tune2
        assert lib.yv3_conv2d_form(d) == 2
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
        y1 = y.clone()
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.isfinite(y).all(), "an output element was not written"
        assert torch.equal(y, y1), "two launches of one descriptor differ"
        return y
    tiles = run(1, 0)
    scale = float(tiles.abs().max())
    for parts in (6, 3, 2):
        y = run(0, parts)
        if parts == 6:
            assert torch.equal(y, tiles), "six one-row parts must add the same products in the same order"
        else:
            assert float((y - tiles).abs().max()) <= 2e-6 * scale, "parts = %d: %g" % (parts, float((y - tiles).abs().max()))
    run(0, 0)                                                # the library's own choice
    assert torch.equal(run(0, 6, _ffi.OPT_WINO4_TILES), tiles)
    assert int(status.item()) == 0
    # the hand-over area is the last 512 * (256 KB + 4) + 256 bytes of the buffer (rounded down to 256); its flags sit in the last 128 KB part
    area_off = (ws.numel() - (512 * (512 * 128 * 4 + 4) + 256)) & ~255
    flags = ws[area_off + 1023 * 128 * 1024: area_off + 1023 * 128 * 1024 + 4 * 1023].view(torch.int32)
    assert int((flags != 0).sum()) == 0, "a hand-over flag was left set"


@pytest.mark.parametrize("cin,cout,B,H,W,k,stride,bn", [(256, 128, 3, 52, 52, 1, 1, True), (512, 256, 5, 26, 26, 1, 1, True), (1024, 512, 2, 13, 13, 1, 1, True),
                                                        (128, 64, 2, 104, 104, 1, 1, True), (96, 192, 1, 37, 23, 1, 1, True), (256, 128, 20, 52, 52, 1, 1, True),
                                                        (64, 64, 1, 5, 3, 1, 1, False), (128, 256, 64, 26, 26, 1, 1, True),
                                                        (64, 128, 3, 52, 52, 3, 2, True), (128, 256, 2, 27, 19, 3, 2, True), (96, 128, 1, 13, 13, 3, 1, True),
                                                        (128, 128, 40, 26, 26, 3, 2, True)])
def test_gemm_f32_equals_tiles_bitwise(cin, cout, B, H, W, k, stride, bn):
    """Exact-fp32 mode: the persistent DMA-fed GEMM (csrc/conv_gemm_f32.hip, round 6) against conv_igemm_f32's tiles (tune[0] = 13) -- same K
    order per output element, so the same bits: with every row on the GEMM (tune[0] = 14, tune[1] = 3), with the library's split (whole
    rounds of the chip on the GEMM, the rest on the tiles) and by its own rule.  Plain 1x1 layers: pixel counts that are not multiples of the
    128 / 256-row tiles (37 x 23, 5 x 3: rows past the last pixel are requested clamped and never stored -- NaN-filled buffer + canary behind
    it), three chunks per tile (cin 96), the 256 x 64 tile (cout 64 / 192), a plain conv (no BN: bias, linear), more than one tile per
    workgroup (20 x 52 x 52: 423 tiles on 256 CUs; 64 x 26 x 26: 676).  3x3 layers (the kernel's K3 operand path, taken with tune[0] = 14 only):
    stride 2 and 1, odd pictures, halo rows and rows past the last pixel zero-filled through the buffer descriptor's bounds.  Against fp64 as the
    direct kernel (2e-5).  yv3_conv2d_launches counts the split."""
    mode = _ffi.F32
    lib = _ffi.lib()
    g = torch.Generator().manual_seed(cin + cout + H + B)
    if bn:
        m = _rand_cbr(cin, cout, k, stride, seed=cin + cout + H).cuda()
    else:
        m = torch.nn.Conv2d(cin, cout, 1, 1, 0, bias=True)
        with torch.no_grad():
            m.weight.copy_(torch.rand(m.weight.shape, generator=g) - 0.5); m.bias.copy_(torch.rand(cout, generator=g) - 0.5)
        m = m.cuda().eval()
    sp = arch.ConvSpec("t", cin, cout, k, stride, bn, False)
    pc = engine.pack_conv(m, sp, mode)
    x = torch.rand(B, H, W, cin, generator=g) * 2 - 0.5
    xg = x.cuda()
    Ho, Wo = engine.out_hw(H, W, k, stride)
    M = B * Ho * Wo

    def run(t0, t1):
        buf = torch.full((M * cout + 4096,), float("nan"), device="cuda")
        y = buf[:M * cout].view(B, Ho, Wo, cout)
        d = engine.make_desc(pc, xg, y, B, H, W, None, dtype=mode)
        d.tune[0], d.tune[1] = t0, t1
        n = lib.yv3_conv2d_launches(d)
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.isfinite(y).all(), "an output element was not written"
        assert torch.isnan(buf[M * cout:]).all(), "a store went past the last pixel"
        return y, n
    tiles, n_tiles = run(13, 0)
    assert n_tiles == 1
    for t0, t1 in ((14, 3), (14, 0), (0, 0)):
        y, n = run(t0, t1)
        assert torch.equal(y, tiles), "tune %d,%d differs from the tiles" % (t0, t1)
        assert n in (1, 2) and (n == 1 or (t1 == 0))
    with torch.no_grad():
        if bn:
            ref = _ref_cbr(m.cpu(), x.permute(0, 3, 1, 2))
        else:
            ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), m.weight.detach().cpu().double(), m.bias.detach().cpu().double())
    assert_close_rel(tiles.permute(0, 3, 1, 2).cpu(), ref, 2e-5, "fp32 conv %s" % ((cin, cout, B, H, W, k, stride),))


def test_eval_letterbox_and_scale_vs_oracle():
    """The evaluation pipeline's input preparation on the GPU (SURVEY 8f-1 eval variant; evaluate.py:211,213): ``IaaLetterbox(dim)``
    (box at ((out - box) // 2): one pixel off the utils rule for odd boxes) and ``iaa.Scale(dim)`` (plain bicubic resize), bit for bit
    against the oracle's restatement (cv2 resampling parity UNPINNED as for yv3_letterbox; geometry pinned by the reference)."""
    from yolo_v3_amd import letterbox_batch, letterbox_transforms
    shapes = [(500, 333), (415, 833), (452, 602), (37, 53), (1080, 1920), (416, 416)]
    imgs = [(synth.uniform01(61 + i, 9, h * w * 3).reshape(h, w, 3) * 255).astype(np.uint8) for i, (h, w) in enumerate(shapes)]
    for dim in ((416, 416), (608, 416)):
        batch, trans = letterbox_batch(imgs, dim, variant="eval")
        for i, im in enumerate(imgs):
            ref = oc.iaa_letterbox_image(im, dim)
            assert torch.equal(batch[i].cpu(), ref), "eval letterbox image %d dim %s: %d values differ" % (i, dim, int((batch[i].cpu() != ref).sum()))
            assert trans[i].tolist()[:4] == list(oc.iaa_letterbox_params(im.shape, dim[1], dim[0]))
        sc, _ = letterbox_batch(imgs, dim, variant="scale")
        for i, im in enumerate(imgs):
            assert torch.equal(sc[i].cpu(), oc.iaa_scale_image(im, dim))
    # the 333 x 500 image: box 277 wide -> x offset 69 here, 70 under utils.letterbox_transforms
    b416, tr = letterbox_batch(imgs[:1], (416, 416), variant="eval")
    assert tr[0].tolist()[:4] == [277, 416, 69, 0] and letterbox_transforms((333, 500), (416, 416))[2] == 70
    assert not torch.equal(b416[0].cpu(), oc.letterbox_image(imgs[0], (416, 416)))
    with pytest.raises(_ffi.Yv3Error):                                     # a box that does not fit the canvas is refused
        _ffi.check(_ffi.lib().yv3_letterbox_ex(b416.data_ptr(), 8, 8, b416.data_ptr(), 416, 416, 417, 10, 0, 0, _ffi.stream_ptr()), "yv3_letterbox_ex")
