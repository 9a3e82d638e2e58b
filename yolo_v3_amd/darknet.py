"""Drop-in surface of the reference's ``darknet.py`` -- model classes + darknet ``.weights`` loader.

Same class names, constructor arguments, attributes and ``state_dict`` keys as the reference
(``/root/reference/darknet.py``), so existing callers (``test.py:35``, ``evaluate.py:201``,
the notebooks) and existing checkpoints work unchanged.  What differs is everything underneath:
the modules below are *parameter containers*; ``forward`` never calls ``nn.Conv2d`` /
``nn.BatchNorm2d`` but hands the whole network to ``engine.Engine``, i.e. to the hand-written
HIP kernels in ``csrc/`` (NHWC implicit-GEMM on MFMA, fused BN/LeakyReLU/residual epilogues,
fused upsample+concat, fused decode).  GPU only: a CPU tensor raises (no fallback).

Not carried over: the training branch (``target is not None`` -> loss, reference
darknet.py:225-229 / yololayer.py:64-95) raises ``NotImplementedError``.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _ffi, arch, engine as _engine
from .yololayer import YoloLayer

DEFAULT_MATH_MODE = _ffi.F32H2

__all__ = ["conv_bn_relu", "res_layer", "Darknet", "PreDetectionConvGroup", "UpsampleGroup",
           "YoloNet", "WeightManager", "map2cfgDict", "make_res_stack"]


def _nchw_to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nhwc_to_nchw(y):
    return y.permute(0, 3, 1, 2).contiguous()


def _run_single(module, spec, x_nhwc, residual=None, x2=None, cin_up=0):
    """One convolution, NHWC in / NHWC out (module-level API; the full net uses engine.Plan)."""
    lib = _ffi.lib()
    _ffi.require_cuda(x_nhwc, "input")
    B, H, W, _ = x_nhwc.shape if x2 is None else x2.shape
    with torch.cuda.device(x_nhwc.device):
        pc = _engine.pack_conv(module, spec, _ffi.F32)
        ho, wo = _engine.out_hw(H, W, spec.k, spec.stride)
        y = torch.empty((B, ho, wo, spec.cout), device=x_nhwc.device, dtype=torch.float32)
        d = _engine.make_desc(pc, x_nhwc, y, B, H, W, residual, x2, cin_up)
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()), "yv3_conv2d")
    return y


class conv_bn_relu(nn.Module):
    """Conv2d(no bias, 'SAME' pad) + BatchNorm2d + LeakyReLU(0.1) (reference darknet.py:27-44)."""

    def __init__(self, nin, nout, ks, s=1, pad='SAME', padding=0, bn=True, act="leakyRelu"):
        super().__init__()
        # The reference's other configurations do not run either: with bn=False its forward calls ``self.bn`` == False (darknet.py:31,43-44:
        # TypeError), with another `act` there is no ``self.relu`` (AttributeError); only an explicit `padding` with pad != 'SAME' works
        # there, and nothing in the reference uses it.
        if pad != 'SAME' or not bn or act != "leakyRelu":
            raise NotImplementedError("only the configuration YoloNet uses is supported (pad='SAME', bn=True, act='leakyRelu'); "
                                      "the reference's bn=False / other-act variants fail in its own forward (darknet.py:31,43-44)")
        self.conv = nn.Conv2d(nin, nout, ks, s, (ks - 1) // 2, bias=False)
        self.bn = nn.BatchNorm2d(nout)
        self.relu = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        self.act = act

    def _spec(self):
        c = self.conv
        return arch.ConvSpec("conv_bn_relu", c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0], True, False)

    def forward_nhwc(self, x_nhwc, residual=None):
        return _run_single(self, self._spec(), x_nhwc, residual)

    def forward(self, x):
        """NCHW in, NCHW out, like the reference module (uses the HIP kernels)."""
        _ffi.require_cuda(x, "input")
        sp = self._spec()
        if sp.cin == 3:
            return _nhwc_to_nchw(_first_conv(self, x))
        if sp.cin % 32:
            raise _ffi.Yv3Error("input channels must be 3 or a multiple of 32")
        return _nhwc_to_nchw(self.forward_nhwc(_nchw_to_nhwc(x.float())))


def _first_conv(module, x_nchw):
    lib = _ffi.lib()
    sp = module._spec()
    if (sp.cin, sp.cout, sp.k, sp.stride) != (3, 32, 3, 1):
        raise _ffi.Yv3Error("3-channel input is only supported for the 3->32 3x3 stride-1 first layer")
    x = x_nchw.float().contiguous()
    B, _, H, W = x.shape
    with torch.cuda.device(x.device):
        pc = _engine.pack_conv(module, sp, _ffi.F32)
        y = torch.empty((B, H, W, 32), device=x.device, dtype=torch.float32)
        _ffi.check(lib.yv3_conv0(x.data_ptr(), pc.w.data_ptr(), pc.alpha.data_ptr(), pc.beta.data_ptr(),
                                 y.data_ptr(), B, H, W, _ffi.F32, None, _ffi.stream_ptr()), "yv3_conv0")
    return y


class res_layer(nn.Module):
    """x + conv3x3(conv1x1(x)) (reference darknet.py:46-53); the add is the second conv's epilogue."""

    def __init__(self, nin):
        super().__init__()
        self.conv1 = conv_bn_relu(nin, nin // 2, ks=1)
        self.conv2 = conv_bn_relu(nin // 2, nin, ks=3)

    def forward_nhwc(self, x):
        return self.conv2.forward_nhwc(self.conv1.forward_nhwc(x), residual=x)

    def forward(self, x):
        return _nhwc_to_nchw(self.forward_nhwc(_nchw_to_nhwc(x.float())))


def map2cfgDict(mlist):
    """darknet-cfg layer index -> position in ``mlist`` (reference darknet.py:55-65): a res_layer
    occupies three cfg entries (1x1, 3x3, shortcut) and is addressed by the last one."""
    table, cfg_idx = {}, 0
    for pos, m in enumerate(mlist):
        if isinstance(m, res_layer):
            table[cfg_idx] = None
            table[cfg_idx + 1] = None
            cfg_idx += 2
        table[cfg_idx] = pos
        cfg_idx += 1
    return table


def make_res_stack(nin, num_blk):
    """Stride-2 down-sampling conv followed by ``num_blk`` residual blocks (reference darknet.py:68-70)."""
    return nn.ModuleList([conv_bn_relu(nin, nin * 2, 3, s=2)] + [res_layer(nin * 2) for _ in range(num_blk)])


class _CachingGroup(nn.Module):
    """The reference's side-channel for routed feature maps (``addCachedOut`` / ``getCachedOut``,
    darknet.py:86-100,122-150), kept for API parity when a group is run on its own."""

    def _run_list(self, x_nhwc):
        for pos, m in enumerate(self.mlist):
            if isinstance(m, nn.Conv2d):
                sp = arch.ConvSpec("head", m.in_channels, m.out_channels, 1, 1, False, False)
                x_nhwc = _run_single(m, sp, x_nhwc)
            else:
                x_nhwc = m.forward_nhwc(x_nhwc)
            if pos in self.cachedOutDict:
                self.cachedOutDict[pos] = _nhwc_to_nchw(x_nhwc)
        return x_nhwc


class Darknet(_CachingGroup):
    """Darknet-53 feature extractor (reference darknet.py:72-104)."""

    def __init__(self, blkList, nout=32):
        super().__init__()
        self.mlist = nn.ModuleList([conv_bn_relu(3, nout, 3)])
        for stage, nb in enumerate(blkList):
            self.mlist += make_res_stack(nout * (2 ** stage), nb)
        self.map2yolocfg = map2cfgDict(self.mlist)
        self.cachedOutDict = dict()

    def forward(self, x):
        _ffi.require_cuda(x, "input")
        y = _first_conv(self.mlist[0], x)
        if 0 in self.cachedOutDict:
            self.cachedOutDict[0] = _nhwc_to_nchw(y)
        for pos in range(1, len(self.mlist)):
            y = self.mlist[pos].forward_nhwc(y)
            if pos in self.cachedOutDict:
                self.cachedOutDict[pos] = _nhwc_to_nchw(y)
        return _nhwc_to_nchw(y)

    def addCachedOut(self, idx, mode="yolocfg"):
        self.cachedOutDict[self.map2yolocfg[idx] if mode == "yolocfg" else idx] = None

    def getCachedOut(self, idx, mode="yolocfg"):
        return self.cachedOutDict[self.map2yolocfg[idx] if mode == "yolocfg" else idx]

    def loadWeight(self, weights_path):
        """Backbone-only darknet file, e.g. darknet53.conv.74 (reference darknet.py:102-104)."""
        return WeightManager(self).loadWeight(weights_path)


class PreDetectionConvGroup(_CachingGroup):
    """(1x1 n, 3x3 2n) x num_conv, then a plain 1x1 conv to 3*(5+numClass) (reference darknet.py:107-150)."""

    def __init__(self, nin, nout, num_conv=3, numClass=80):
        super().__init__()
        self.mlist = nn.ModuleList()
        for i in range(num_conv):
            self.mlist += [conv_bn_relu(nin, nout, ks=1), conv_bn_relu(nout, nout * 2, ks=3)]
            nin = nout * 2
        self.mlist += [nn.Conv2d(nin, (numClass + 5) * 3, 1)]
        self.map2yolocfg = map2cfgDict(self.mlist)
        self.cachedOutDict = dict()

    def forward(self, x):
        return _nhwc_to_nchw(self._run_list(_nchw_to_nhwc(_ffi.require_cuda(x, "input").float())))

    def getIdxFromYoloIdx(self, idx):
        return len(self.map2yolocfg) + idx if idx < 0 else self.map2yolocfg[idx]

    def _resolve(self, idx, mode):
        if mode == "yolocfg":
            return self.getIdxFromYoloIdx(idx)
        return len(self.mlist) - idx if idx < 0 else idx      # (sic) the reference's formula, darknet.py:134

    def addCachedOut(self, idx, mode="yolocfg"):
        self.cachedOutDict[self._resolve(idx, mode)] = None

    def getCachedOut(self, idx, mode="yolocfg"):
        return self.cachedOutDict[self._resolve(idx, mode)]


class UpsampleGroup(nn.Module):
    """1x1 conv_bn_relu, nearest x2, cat((up, route_tail), 1) (reference darknet.py:153-162)."""

    def __init__(self, nin):
        super().__init__()
        self.conv = conv_bn_relu(nin, nin // 2, ks=1)

    def forward(self, route_head, route_tail):
        _ffi.require_cuda(route_head, "route_head")
        _ffi.require_cuda(route_tail, "route_tail")
        up = self.conv.forward(route_head)
        # stand-alone use only (inside YoloNet the upsample + concat is folded into the consumer conv's gather): one HIP launch
        # (csrc/prepost.hip: yv3_upsample2x_concat), no vendor elementwise ops
        B, cu, h, w = up.shape
        tail = route_tail.float().contiguous()
        if tuple(tail.shape[0:1] + tail.shape[2:]) != (B, 2 * h, 2 * w):
            raise _ffi.Yv3Error("route_tail must be [B, C, 2h, 2w] for route_head [B, C', h, w]: got %s and %s"
                                % (tuple(route_tail.shape), tuple(route_head.shape)))
        out = torch.empty((B, cu + tail.shape[1], 2 * h, 2 * w), device=up.device, dtype=torch.float32)
        with torch.cuda.device(up.device):
            _ffi.check(_ffi.lib().yv3_upsample2x_concat(up.data_ptr(), tail.data_ptr(), out.data_ptr(), B, cu, tail.shape[1], h, w,
                                                        _ffi.stream_ptr()), "yv3_upsample2x_concat")
        return out


class YoloNet(nn.Module):
    """YOLOv3 (reference darknet.py:167-246).  ``forward(x)`` -> ``(det1, det2, det3)``."""

    def __init__(self, img_dim, anchors=list(arch.DEFAULT_ANCHORS), numClass=80):
        super().__init__()
        self.numClass = numClass
        self.img_dim = img_dim
        self.anchors_flat = list(anchors)
        self.stat_keys = ['loss', 'loss_x', 'loss_y', 'loss_w', 'loss_h', 'loss_conf', 'loss_cls',
                          'nCorrect', 'nGT', 'recall']
        pairs = [(anchors[i], anchors[i + 1]) for i in range(0, len(anchors), 2)]

        self.feature = Darknet(list(arch.BACKBONE_BLOCKS))
        self.feature.addCachedOut(61)
        self.feature.addCachedOut(36)

        self.pre_det1 = PreDetectionConvGroup(1024, 512, numClass=numClass)
        self.yolo1 = YoloLayer(pairs, list(arch.ANCHOR_MASKS[0]), img_dim, numClass)
        self.pre_det1.addCachedOut(-3)

        self.up1 = UpsampleGroup(512)
        self.pre_det2 = PreDetectionConvGroup(768, 256, numClass=numClass)
        self.yolo2 = YoloLayer(pairs, list(arch.ANCHOR_MASKS[1]), img_dim, numClass)
        self.pre_det2.addCachedOut(-3)

        self.up2 = UpsampleGroup(256)
        self.pre_det3 = PreDetectionConvGroup(384, 128, numClass=numClass)
        self.yolo3 = YoloLayer(pairs, list(arch.ANCHOR_MASKS[2]), img_dim, numClass)

        self._engines = {}
        # Convolution math mode (include/yv3.h).  All three fp32 modes meet the 1e-4 parity bar against the
        # reference (tests/test_gpu_e2e.py):
        #   F32H2 (default) fp16 hi+lo planes, 3 fp16 MFMAs per fp32 product   -- fastest; activations must stay
        #                   within +-65504 (checked: a saturated value raises Yv3Error instead of passing silently)
        #   F32X3           exact 3-way bf16 split, 6 bf16 MFMAs per product   -- fp32 exponent range, ~1.6x slower
        #   F32             exact fp32 MFMA                                    -- ~2.7x slower
        # _ffi.BF16 is the reduced-precision throughput mode (not a 1e-4 mode).
        self.math_mode = DEFAULT_MATH_MODE

    # ---- HIP execution
    def engine(self, dtype=None):
        dtype = self.math_mode if dtype is None else dtype
        # a network whose activations once left the fp16 range of F32H2 runs in the fall-back mode from then on (engine.range_fallback;
        # ``net.strict_range = True``: raise instead)
        if dtype == _engine.F32H2 and self.__dict__.get("_range_fallback") is not None and not getattr(self, "strict_range", False):
            dtype = self._range_fallback
        eng = self._engines.get(dtype)
        if eng is None:
            eng = self._engines[dtype] = _engine.Engine(self, dtype)
        return eng

    def repack(self):
        """Invalidate packed weights and plans IN PLACE; the next forward re-packs from the current parameters.  A
        `Detector` (or engine) the caller already holds sees the change too: it keeps its Engine object, whose
        generation counter moves with the re-pack.

        Packed weights follow the parameters automatically for everything that changes a parameter's
        ``(data_ptr, _version)``: ``load_state_dict``, ``loadWeight`` / ``WeightManager``, assignments, in-place ops on
        the parameter.  Writes through ``param.data`` (``p.data.copy_(..)``, the reference loader's idiom,
        darknet.py:275) are invisible to that check: call ``repack()`` after them, or set
        ``net.weight_check = "checksum"`` (engine.Engine._signature)."""
        for eng in self._engines.values():
            eng.invalidate()

    def load_state_dict(self, state_dict, *args, **kwargs):
        out = super().load_state_dict(state_dict, *args, **kwargs)
        self.repack()
        return out

    # engines / detectors hold ctypes descriptor arrays (raw device pointers): never copied or pickled with the module
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engines"] = {}
        state.pop("_range_fallback", None)
        state.pop("_detectors", None)
        state.pop("_sharded_detectors", None)
        return state

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def forward_cat(self, x, dtype=None):
        """The three scales already concatenated: ``[B, N, 5+C]`` == ``torch.cat((det1,det2,det3), 1)``."""
        dets, _ = self.engine(dtype).forward(x)
        return dets

    def forward(self, x, target=None):
        if target is not None:
            raise NotImplementedError("training loss (reference darknet.py:225-229, yololayer.py:64-95) "
                                      "is outside the inference hot path")
        dets, plan = self.engine().forward(x)
        r1, r2 = plan.rows[0], plan.rows[0] + plan.rows[1]
        return dets[:, :r1], dets[:, r1:r2], dets[:, r2:]

    # ---- weights (reference darknet.py:234-246)
    def saveWeight(self, weights_path, format='pytorch'):
        if format == 'pytorch':
            torch.save(self.state_dict(), weights_path)
        elif format == 'darknet':
            # the reference raises NotImplementedError here (darknet.py:237-238); writing the file completes
            # the wire format both ways
            WeightManager(self).saveWeight(weights_path)

    def loadWeight(self, weights_path, format='pytorch'):
        if format == 'pytorch':
            self.load_state_dict(torch.load(weights_path, map_location="cpu"))
        elif format == 'darknet':
            WeightManager(self).loadWeight(weights_path)
        self.repack()


class WeightManager:
    """Reader for darknet ``.weights`` files (reference darknet.py:249-303).

    File = 5 x int32 header (major, minor, revision, seen-lo, seen-hi) + float32 stream.  Per
    ``conv_bn_relu``: bn.bias, bn.weight, running_mean, running_var, conv.weight [cout,cin,k,k];
    per plain conv: bias, weight.  Convs are visited in module pre-order (== darknet cfg order).
    """

    def __init__(self, model):
        self.model = model
        self.conv_list = self.find_conv_layers(model)
        self.header = None
        self.seen = None

    @staticmethod
    def find_conv_layers(mod):
        found, inside = [], set()
        for m in mod.modules():
            if isinstance(m, conv_bn_relu):
                found.append(m)
                inside.add(id(m.conv))
            elif isinstance(m, nn.Conv2d) and id(m) not in inside:
                found.append(m)
        return found

    def read_file(self, file):
        with open(file, "rb") as fp:
            header = np.fromfile(fp, dtype=np.int32, count=5)
            self.header = torch.from_numpy(header)
            self.seen = self.header[3]
            return np.fromfile(fp, dtype=np.float32)

    def loadWeight(self, weight_path):
        stream = self.read_file(weight_path)
        return self.load_stream(stream)

    def to_stream(self):
        """The parameters as one float32 numpy stream in darknet file order (inverse of load_stream)."""
        parts = []
        for m in self.conv_list:
            ts = (m.bn.bias, m.bn.weight, m.bn.running_mean, m.bn.running_var, m.conv.weight) if isinstance(m, conv_bn_relu) \
                else (m.bias, m.weight)
            parts += [t.detach().float().cpu().contiguous().numpy().ravel() for t in ts]
        return np.concatenate(parts).astype(np.float32)

    def saveWeight(self, weight_path, seen=None):
        """Write a darknet ``.weights`` file: int32 (major=0, minor=2, revision=0), int64 ``seen``, float32 stream.
        ``seen`` defaults to the value read by the last loadWeight (0 if none)."""
        if seen is None:
            seen = int(self.seen) if self.seen is not None else 0
        with open(weight_path, "wb") as fp:
            np.array([0, 2, 0], dtype=np.int32).tofile(fp)
            np.array([seen], dtype=np.int64).tofile(fp)
            self.to_stream().tofile(fp)

    def load_stream(self, stream):
        """Consume a float32 stream (numpy) already in memory; returns the number of floats used."""
        ptr = 0

        def take(param):
            nonlocal ptr
            n = param.numel()
            if ptr + n > stream.size:
                raise ValueError("weight stream too short: need %d floats, have %d" % (ptr + n, stream.size))
            with torch.no_grad():
                param.copy_(torch.from_numpy(stream[ptr:ptr + n].copy()).view_as(param))
            ptr += n

        for m in self.conv_list:
            if isinstance(m, conv_bn_relu):
                for t in (m.bn.bias, m.bn.weight, m.bn.running_mean, m.bn.running_var, m.conv.weight):
                    take(t)
            else:
                take(m.bias)
                take(m.weight)
        if hasattr(self.model, "repack"):
            self.model.repack()
        return ptr
