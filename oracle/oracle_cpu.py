"""CPU oracle for the YOLOv3 inference hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (PyTorch fp32 CPU ops + numpy) of the reference
algorithm for the path  Darknet-53 forward -> YOLO head decode -> IOU + greedy NMS.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / the reported CPU baseline.  Nothing in
``yolo_v3_amd/`` imports it; the product path has no CPU fallback.

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself,
produced in the build container by ``oracle/make_golden.py`` (which imports
``/root/reference``) and committed as fixtures under ``tests/golden/``;
``tests/test_oracle_golden.py`` replays them.

It is written as pure functions over a ``state_dict`` (reference key names) rather
than as nn.Modules, citing the reference lines each function follows.
"""
import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_ANCHORS = [10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326]
BLOCKS = (1, 2, 8, 8, 4)


# --------------------------------------------------------------------------- weights
def read_darknet_file(path):
    """reference darknet.py:265-271: 5 x int32 header, then a float32 stream."""
    with open(path, "rb") as fp:
        header = np.fromfile(fp, dtype=np.int32, count=5)
        stream = np.fromfile(fp, dtype=np.float32)
    return header, stream


def _cbr_keys(prefix):
    return [prefix + ".bn.bias", prefix + ".bn.weight", prefix + ".bn.running_mean",
            prefix + ".bn.running_var", prefix + ".conv.weight"]


def conv_prefixes(num_class=80):
    """(prefix, cin, cout, k, has_bn) in the order reference darknet.py:292-303 visits the convs."""
    out = [("feature.mlist.0", 3, 32, 3, True)]
    idx, c = 1, 32
    for nblk in BLOCKS:
        out.append(("feature.mlist.%d" % idx, c, 2 * c, 3, True)); idx += 1
        c *= 2
        for _ in range(nblk):
            out.append(("feature.mlist.%d.conv1" % idx, c, c // 2, 1, True))
            out.append(("feature.mlist.%d.conv2" % idx, c // 2, c, 3, True))
            idx += 1
    nout_head = 3 * (5 + num_class)

    def branch(pre, nin, n):
        cin = nin
        for i in range(3):
            out.append(("%s.mlist.%d" % (pre, 2 * i), cin, n, 1, True))
            out.append(("%s.mlist.%d" % (pre, 2 * i + 1), n, 2 * n, 3, True))
            cin = 2 * n
        out.append(("%s.mlist.6" % pre, cin, nout_head, 1, False))

    branch("pre_det1", 1024, 512)
    out.append(("up1.conv", 512, 256, 1, True))
    branch("pre_det2", 768, 256)
    out.append(("up2.conv", 256, 128, 1, True))
    branch("pre_det3", 384, 128)
    return out


def state_dict_from_stream(stream, num_class=80):
    """reference darknet.py:254-290: slice the float stream into parameters, cfg order."""
    sd, p = {}, 0
    for prefix, cin, cout, k, has_bn in conv_prefixes(num_class):
        nw = cout * cin * k * k
        if has_bn:
            for key in _cbr_keys(prefix)[:4]:
                sd[key] = torch.from_numpy(stream[p:p + cout].copy()); p += cout
            sd[prefix + ".conv.weight"] = torch.from_numpy(stream[p:p + nw].copy()).view(cout, cin, k, k); p += nw
        else:
            sd[prefix + ".bias"] = torch.from_numpy(stream[p:p + cout].copy()); p += cout
            sd[prefix + ".weight"] = torch.from_numpy(stream[p:p + nw].copy()).view(cout, cin, k, k); p += nw
    return sd, p


# --------------------------------------------------------------------------- conv trunk
# ``prec`` selects the arithmetic the conv trunk is restated in:
#   None    the reference as it runs: fp32 everywhere (darknet.py:43-44,52-53,118)
#   "bf16"  BASELINE configs[2] "bf16 convs / fp32 decode": the SAME reference ops with every conv's operands
#           rounded to bfloat16 (round-to-nearest-even) -- the weights of every conv except the 3-channel first
#           layer, and every STORED activation, i.e. the output of each conv_bn_relu (after BN + LeakyReLU,
#           darknet.py:43-44), of each res_layer (after the fp32 add, darknet.py:52-53) and of UpsampleGroup's
#           conv -- with fp32 accumulation, fp32 BN/activation/residual add, fp32 head logits (darknet.py:118),
#           fp32 decode and post-processing.  A bf16 x bf16 product is exact in fp32, so the only difference
#           between two implementations of this definition is the fp32 summation order.  Pinned by running the
#           reference's own modules with rounding hooks (oracle/make_golden_bf16.py -> tests/golden/e2e_bf16.npz).
#   "bf16-halves"  the same definition evaluated in ANOTHER fp32 summation order (each conv as the sum of two convs over
#           the two halves of its input channels): tests use the spread between the two CPU evaluations as the yardstick
#           for what any implementation of the definition can be held to end to end.
def round_bf16(t):
    """fp32 -> nearest bfloat16 (ties to even) -> fp32."""
    return t.bfloat16().float()


def cbr(sd, prefix, x, stride=1, prec=None, store=True):
    """reference darknet.py:27-44: conv(no bias, pad=(k-1)//2) -> BatchNorm2d(eval) -> LeakyReLU(0.1)."""
    w = sd[prefix + ".conv.weight"]
    bf = prec in ("bf16", "bf16-halves")
    if bf and w.shape[1] != 3:
        w = round_bf16(w)
    pad = (w.shape[2] - 1) // 2
    if prec == "bf16-halves" and w.shape[1] >= 2:
        h = w.shape[1] // 2
        y = F.conv2d(x[:, :h], w[:, :h], None, stride, pad) + F.conv2d(x[:, h:], w[:, h:], None, stride, pad)
    else:
        y = F.conv2d(x, w, None, stride, pad)
    y = F.batch_norm(y, sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"],
                     sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"], False, 0.1, 1e-5)
    y = F.leaky_relu(y, 0.1)
    return round_bf16(y) if (bf and store) else y


def backbone(sd, x, taps=None, prec=None):
    """reference darknet.py:72-88 + 46-53.  Returns (out, route36 [52x52x256], route61 [26x26x512])."""
    x = cbr(sd, "feature.mlist.0", x, prec=prec)
    if taps is not None: taps.append(("feature.mlist.0", x))
    idx, routes = 1, {}
    for nblk in BLOCKS:
        x = cbr(sd, "feature.mlist.%d" % idx, x, stride=2, prec=prec)
        if taps is not None: taps.append(("feature.mlist.%d" % idx, x))
        idx += 1
        for _ in range(nblk):
            p = "feature.mlist.%d" % idx
            h = cbr(sd, p + ".conv1", x, prec=prec)
            if taps is not None: taps.append((p + ".conv1", h))
            x = x + cbr(sd, p + ".conv2", h, prec=prec, store=False)   # darknet.py:53 (the sum is what is stored)
            if prec in ("bf16", "bf16-halves"):
                x = round_bf16(x)
            if taps is not None: taps.append((p + ".conv2", x))
            idx += 1
        routes[idx - 1] = x
    # darknet.py:180-181: cfg idx 61 -> mlist[23], cfg idx 36 -> mlist[14]
    return x, routes[14], routes[23]


def predet(sd, prefix, x, taps=None, prec=None):
    """reference darknet.py:107-127.  Returns (head logits NCHW, output of mlist[4])."""
    route = None
    for i in range(6):
        x = cbr(sd, "%s.mlist.%d" % (prefix, i), x, prec=prec)
        if taps is not None: taps.append(("%s.mlist.%d" % (prefix, i), x))
        if i == 4:
            route = x                                               # darknet.py:185 addCachedOut(-3)
    w = sd[prefix + ".mlist.6.weight"]
    logits = F.conv2d(x, round_bf16(w) if prec in ("bf16", "bf16-halves") else w, sd[prefix + ".mlist.6.bias"])   # fp32 logits
    if taps is not None: taps.append((prefix + ".mlist.6", logits))
    return logits, route


def upsample_cat(sd, prefix, head, tail, taps=None, prec=None):
    """reference darknet.py:159-162: 1x1 cbr, nearest x2, cat((up, tail), dim=1)."""
    out = cbr(sd, prefix + ".conv", head, prec=prec)
    if taps is not None: taps.append((prefix + ".conv", out))
    out = F.interpolate(out, scale_factor=2, mode="nearest")
    return torch.cat((out, tail), 1)


def head_logits(sd, x, taps=None, prec=None):
    """Conv trunk only: the three head logit maps [B,255,h,w] (reference darknet.py:198-223)."""
    feat, r36, r61 = backbone(sd, x, taps, prec)
    l1, h1 = predet(sd, "pre_det1", feat, taps, prec)
    l2, h2 = predet(sd, "pre_det2", upsample_cat(sd, "up1", h1, r61, taps, prec), taps, prec)
    l3, _ = predet(sd, "pre_det3", upsample_cat(sd, "up2", h2, r36, taps, prec), taps, prec)
    return l1, l2, l3


# --------------------------------------------------------------------------- decode
def decode(x, anchors_all, anchors_mask, img_dim, num_class=80):
    """reference yololayer.py:31-59,97-105 (inference branch), same op order.

    x [B, 3*(5+C), H, W] -> [B, H*W*3, 5+C]; row = (y*W + x)*3 + a; cols cx,cy,w,h,conf,cls...
    """
    nB, nA = x.shape[0], len(anchors_mask)
    nH, nW = x.shape[2], x.shape[3]
    attrib = 5 + num_class
    stride = img_dim[1] / nH                                         # yololayer.py:36 (python float)
    pairs = [(anchors_all[i], anchors_all[i + 1]) for i in range(0, len(anchors_all), 2)] \
        if not isinstance(anchors_all[0], (tuple, list)) else list(anchors_all)
    anc_all = torch.FloatTensor(pairs) / stride                      # :37
    anc = anc_all[list(anchors_mask)]                                # :38
    preds = x.view(nB, nA, attrib, nH, nW).permute(0, 1, 3, 4, 2).contiguous()   # :42
    xy = preds[..., :2].sigmoid()                                    # :45
    wh = preds[..., 2:4]
    conf = preds[..., 4].sigmoid()                                   # :47
    cls = preds[..., 5:].sigmoid()                                   # :48
    gx = torch.arange(nW).repeat(nH, 1).unsqueeze(2)                 # :51
    gy = torch.arange(nH).repeat(nW, 1).t().unsqueeze(2)             # :52
    gxy = torch.cat((gx, gy), 2).float()
    manc = anc.view(1, nA, 1, 1, 2).repeat(1, 1, nH, nW, 1)
    box = torch.empty(preds[..., :4].shape)
    box[..., :2] = xy + gxy                                          # :58
    box[..., 2:4] = wh.exp() * manc                                  # :59
    out = torch.cat((box * stride, conf.unsqueeze(4), cls), 4)       # :98-100
    return out.permute(0, 2, 3, 1, 4).contiguous().view(nB, nA * nH * nW, attrib)   # :104


def yolonet_forward(sd, x, anchors=DEFAULT_ANCHORS, num_class=80, prec=None):
    """reference darknet.py:198-231 with target=None: returns (det1, det2, det3)."""
    img_dim = (x.shape[3], x.shape[2])                               # darknet.py:199
    l1, l2, l3 = head_logits(sd, x, prec=prec)
    return (decode(l1, anchors, (6, 7, 8), img_dim, num_class),
            decode(l2, anchors, (3, 4, 5), img_dim, num_class),
            decode(l3, anchors, (0, 1, 2), img_dim, num_class))


# --------------------------------------------------------------------------- geometry
def cxcywh_to_x1y1x2y2(box):
    """reference boundingbox.py:25-29 (returns a new tensor)."""
    x1, x2 = box[..., 0] - box[..., 2] / 2, box[..., 0] + box[..., 2] / 2
    y1, y2 = box[..., 1] - box[..., 3] / 2, box[..., 1] + box[..., 3] / 2
    return torch.stack((x1, y1, x2, y2), -1)


def iou_matrix(b):
    """reference utils.py:98-119: all-pairs IOU of x1y1x2y2 boxes, no +1, no eps."""
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ix1 = torch.max(x1[:, None], x1[None, :])
    iy1 = torch.max(y1[:, None], y1[None, :])
    ix2 = torch.min(x2[:, None], x2[None, :])
    iy2 = torch.min(y2[:, None], y2[None, :])
    inter = torch.clamp(ix2 - ix1, min=0) * torch.clamp(iy2 - iy1, min=0)
    area = (x2 - x1) * (y2 - y1)
    union = area[None, :] + area[:, None] - inter                    # utils.py:116 (area_j + area_i) - inter
    return inter / union


def bbox_iou(b1, b2, mode="x1y1x2y2"):
    """reference utils.py:122-146: rectangular IOU [n1,n2], two box formats."""
    if mode == "cxcywh":
        b1 = cxcywh_to_x1y1x2y2(b1)
        b2 = cxcywh_to_x1y1x2y2(b2)
    ix1 = torch.max(b1[:, None, 0], b2[None, :, 0])
    iy1 = torch.max(b1[:, None, 1], b2[None, :, 1])
    ix2 = torch.min(b1[:, None, 2], b2[None, :, 2])
    iy2 = torch.min(b1[:, None, 3], b2[None, :, 3])
    inter = torch.clamp(ix2 - ix1, min=0) * torch.clamp(iy2 - iy1, min=0)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    return inter / (a1[:, None] + a2[None, :] - inter)


# --------------------------------------------------------------------------- post-processing
def _greedy_keep(over):
    """Greedy scan of reference utils.py:180-190 on a boolean [n,n] matrix (True = IOU > thr).

    A box whose diagonal entry is False (zero-area -> NaN self-IOU, or thr >= 1) is skipped:
    it is neither emitted nor allowed to suppress (utils.py:182).
    """
    over = over.numpy().copy()
    n = over.shape[0]
    alive = over.diagonal().copy()
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        sup = over[i, i + 1:] & alive[i + 1:]
        alive[i + 1:][sup] = False
    return keep


def postprocess(detections, num_classes, obj_conf_thr=0.5, nms_thr=0.4, is_eval=False, use_nms=True):
    """reference utils.py:226-258 + 148-224.  Does not mutate its argument (the reference does
    when handed a CPU tensor, utils.py:227-233).  Returns ``[]`` or a list of B tensors, each
    ``[n,7] = x1,y1,x2,y2,conf,score,cls`` or an empty ``torch.Tensor()`` (shape (0,)).
    """
    det = detections.detach().cpu().float().clone()
    det[..., :4] = cxcywh_to_x1y1x2y2(det[..., :4])                  # utils.py:230
    det[..., 5:5 + num_classes] = det[..., 5:5 + num_classes] * det[..., 4].unsqueeze(-1)   # :233
    scores = det[..., 5:5 + num_classes]
    if is_eval:
        index = (scores > obj_conf_thr).nonzero()                    # :238
    else:
        mx, arg = torch.max(scores, -1)                              # :242 (first index wins ties)
        m = mx > obj_conf_thr
        if not m.any():
            return []                                                # :248
        index = torch.cat((m.nonzero(), arg[m].unsqueeze(-1)), -1)
    if len(index) == 0:
        return []                                                    # :251
    results = []
    for b in range(det.shape[0]):
        sel = index[index[:, 0] == b]
        if len(sel) == 0:
            results.append(torch.Tensor())                           # :153-158
            continue
        if not use_nms:                                              # utils.py:204-224
            rows = det[b, sel[:, 1]]
            prob = rows[torch.arange(len(sel)), 5 + sel[:, 2]]
            results.append(torch.cat((rows[:, :5], prob[:, None], sel[:, 2].float()[:, None]), -1))
            continue
        parts = []
        for c in sel[:, 2].unique():                                 # ascending class ids, utils.py:161
            rows = det[b, sel[sel[:, 2] == c][:, 1]]
            order = torch.sort(rows[:, 5 + c], descending=True, stable=True)[1]   # :171 (stable here)
            rows = rows[order]
            keep = _greedy_keep(iou_matrix(rows[:, :4]) > nms_thr)   # :175-190
            rows = rows[keep].view(-1, 5 + num_classes)
            parts.append(torch.cat((rows[:, :5], rows[:, 5 + c].view(-1, 1),
                                    torch.full((len(rows), 1), float(c))), -1))    # :193-197
        results.append(torch.cat(parts, 0) if parts else torch.Tensor())
    return results


def detect(sd, imgs, num_classes=80, obj_conf_thr=0.5, nms_thr=0.4, is_eval=False, use_nms=True,
           anchors=DEFAULT_ANCHORS, prec=None):
    """The caller idiom of reference test.py:35-36 / evaluate.py:201-204."""
    with torch.no_grad():
        d1, d2, d3 = yolonet_forward(sd, imgs, anchors, num_classes, prec)
        return postprocess(torch.cat((d1, d2, d3), 1), num_classes, obj_conf_thr, nms_thr, is_eval, use_nms)


# --------------------------------------------------------------------------- neighbours of the path
def letterbox_transforms(inner_dim, outer_dim):
    """reference utils.py:34-42."""
    ow, oh = outer_dim
    iw, ih = inner_dim
    ratio = min(ow / iw, oh / ih)
    bw, bh = int(iw * ratio), int(ih * ratio)
    return bw, bh, (ow // 2) - (bw // 2), (oh // 2) - (bh // 2), ratio


# cv2.resize for 8-bit images, restated.  THIRD-PARTY ALGORITHM, ABSENT HERE: the reference calls OpenCV
# (``cv2.resize(img, (box_w, box_h), interpolation=cv2.INTER_CUBIC)`` utils.py:50 and ``cv2.resize(img, dim)``
# utils.py:68-69); cv2 is not installed in this image and OpenCV is not vendored by the reference (its README pins
# no version; any 3.4/4.x ``modules/imgproc/src/resize.cpp`` has the algorithm below).  What is restated is the
# published fixed-point path of ``cv::resize`` for CV_8U:
#   * sample position   fx = (float)((dx + 0.5) * scale_x - 0.5), scale_x = 1 / ((double)dst_w / src_w);
#                       sx = floor(fx); fx -= sx                       (resize.cpp, cv::hal::resize coordinate tables)
#   * INTER_CUBIC       coefficients ``interpolateCubic`` (A = -0.75, float32 arithmetic in the written order),
#                       stored as ``saturate_cast<short>(c * 2048)`` (INTER_RESIZE_COEF_BITS = 11, round-half-even);
#                       horizontal pass ``HResizeCubic<uchar,int,short>``: int32 sums of 4 taps, replicated border;
#                       vertical pass ``VResizeCubic`` + ``FixedPtCast<int,uchar,22>``:
#                       ``saturate_cast<uchar>((sum4 + (1 << 21)) >> 22)``
#   * INTER_LINEAR      (the default of ``cv2.resize(img, dim)``): coefficients (1 - fx, fx) * 2048 as shorts, left /
#                       right edge clamps (sx < 0 -> sx = 0, fx = 0; sx >= w - 1 -> sx = w - 1, fx = 0); vertical pass
#                       ``VResizeLinear<uchar,int,short,...>``:
#                       ``uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)``;
#                       an exact 2x2 down-scale is rerouted to INTER_AREA: ``(a + b + c + d + 2) >> 2``.
# PARITY UNPINNED: without cv2 no golden vector can be produced here.  Known caveat: SIMD builds of OpenCV run the
# cubic vertical pass of the bulk of each row in float32 (``VResizeCubicVec_32s8u``: v_muladd + v_round), which can
# differ from this scalar fixed-point definition by 1 LSB on ~1e-4 of the pixels (exact and near ties).
INTER_RESIZE_COEF_SCALE = 2048


def _cv_coords(dst_n, src_n):
    """(integer source index, float32 fraction) per destination index, as cv::resize tabulates them."""
    scale = 1.0 / (float(dst_n) / float(src_n))                                   # double
    f = ((np.arange(dst_n, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    return s, (f - s.astype(np.float32)).astype(np.float32)


def _cv_round_short(c):
    """saturate_cast<short>(float): round half to even, clamp."""
    return np.clip(np.rint(c.astype(np.float32) * np.float32(INTER_RESIZE_COEF_SCALE)), -32768, 32767).astype(np.int64)


def _cv_cubic_coeffs(x):
    """interpolateCubic (float32, operation order of the source)."""
    x = x.astype(np.float32)
    A = np.float32(-0.75)
    one, x1 = np.float32(1.0), (x + np.float32(1.0)).astype(np.float32)
    c0 = ((A * x1 - np.float32(5.0) * A) * x1 + np.float32(8.0) * A) * x1 - np.float32(4.0) * A
    c1 = ((A + np.float32(2.0)) * x - (A + np.float32(3.0))) * x * x + one
    xm = (one - x).astype(np.float32)
    c2 = ((A + np.float32(2.0)) * xm - (A + np.float32(3.0))) * xm * xm + one
    c3 = one - c0 - c1 - c2
    return np.stack([c.astype(np.float32) for c in (c0, c1, c2, c3)], -1)


def cv_resize_cubic_u8(img, dst_w, dst_h):
    """cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_CUBIC) for uint8 [H,W,C] (fixed-point path)."""
    H, W = img.shape[:2]
    sx, fx = _cv_coords(dst_w, W)
    sy, fy = _cv_coords(dst_h, H)
    ax, ay = _cv_round_short(_cv_cubic_coeffs(fx)), _cv_round_short(_cv_cubic_coeffs(fy))   # [dst,4] int
    cols = np.clip(sx[:, None] + np.arange(-1, 3)[None, :], 0, W - 1)                      # replicate border
    rows = np.clip(sy[:, None] + np.arange(-1, 3)[None, :], 0, H - 1)
    src = img.astype(np.int64)
    hor = (src[:, cols, :] * ax[None, :, :, None]).sum(2)                                  # [H,dst_w,C] int32 range
    val = (hor[rows, :, :] * ay[:, :, None, None]).sum(1)                                  # [dst_h,dst_w,C]
    return np.clip((val + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def cv_resize_linear_u8(img, dst_w, dst_h):
    """cv2.resize(img, (dst_w, dst_h)) -- the default INTER_LINEAR -- for uint8 [H,W,C] (fixed-point path)."""
    H, W = img.shape[:2]
    if W == 2 * dst_w and H == 2 * dst_h:                                                   # rerouted to INTER_AREA (resizeAreaFast)
        s = img.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, fx = _cv_coords(dst_w, W)
    sy, fy = _cv_coords(dst_h, H)
    lo, hi = sx < 0, sx >= W - 1
    fx = np.where(lo | hi, np.float32(0.0), fx).astype(np.float32)
    sx = np.where(lo, 0, np.where(hi, W - 1, sx))
    ax = _cv_round_short(np.stack((np.float32(1.0) - fx, fx), -1))
    ay = _cv_round_short(np.stack((np.float32(1.0) - fy, fy), -1))
    cols = np.clip(sx[:, None] + np.arange(2)[None, :], 0, W - 1)
    rows = np.clip(sy[:, None] + np.arange(2)[None, :], 0, H - 1)
    src = img.astype(np.int64)
    hor = (src[:, cols, :] * ax[None, :, :, None]).sum(2)
    r0, r1 = hor[rows[:, 0]], hor[rows[:, 1]]
    b0, b1 = ay[:, 0, None, None], ay[:, 1, None, None]
    return (((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2)).astype(np.uint8)


def letterbox_image(img, dim):
    """reference utils.py:44-72 (letterbox_image + the /255, CHW of load_image): cv2.INTER_CUBIC resize to the box
    (restated above, parity with cv2 UNPINNED), pasted on a 128-grey canvas.  img: uint8 [H,W,3]; dim = (w,h).
    Returns float32 [3,h,w] in [0,1]."""
    H, W = img.shape[:2]
    ow, oh = dim
    bw, bh, bx, by, _ = letterbox_transforms((W, H), (ow, oh))
    canvas = np.full((oh, ow, 3), 128, dtype=np.uint8)
    canvas[by:by + bh, bx:bx + bw] = cv_resize_cubic_u8(img, bw, bh)
    return torch.from_numpy((canvas.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1).copy())


def iaa_letterbox_params(image_shape, new_h, new_w):
    """reference transforms.py:196-205 (IaaLetterbox._compute_height_width_pad): the EVALUATION pipeline's letterbox geometry --
    (resize_w, resize_h, x_pad, y_pad) with the pads floored from (new - resized) / 2 (utils.letterbox_transforms uses
    new // 2 - resized // 2: one pixel more when `new` is even and `resized` odd)."""
    img_h, img_w = image_shape[0:2]
    ratio = min(new_w / img_w, new_h / img_h)
    resize_w, resize_h = int(img_w * ratio), int(img_h * ratio)
    return resize_w, resize_h, (new_w - resize_w) // 2, (new_h - resize_h) // 2


def iaa_letterbox_image(img, dim):
    """evaluate.py:211 `Compose([IaaAugmentations([IaaLetterbox(dim)]), ToTensor()])` for one image (transforms.py:144-172,25-43):
    imgaug's `imresize_single_image(..., interpolation="cubic")` is cv2.resize(INTER_CUBIC) (third-party, absent: restated above,
    parity with cv2 UNPINNED), np.pad with 128, /255, CHW.  img: uint8 [H,W,3]; dim = (w,h)."""
    rw, rh, xp, yp = iaa_letterbox_params(img.shape, dim[1], dim[0])
    canvas = np.full((dim[1], dim[0], 3), 128, dtype=np.uint8)
    canvas[yp:yp + rh, xp:xp + rw] = cv_resize_cubic_u8(img, rw, rh)
    return torch.from_numpy((canvas.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1).copy())


def iaa_scale_image(img, dim):
    """evaluate.py:213 `iaa.Scale(dim)` + ToTensor: imgaug's default interpolation is "cubic" -> cv2.resize(img, dim, INTER_CUBIC)."""
    out = cv_resize_cubic_u8(img, dim[0], dim[1])
    return torch.from_numpy((out.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1).copy())


def resize_image(img, dim):
    """reference utils.py:68-71, mode='resize': cv2.resize(img, dim) (INTER_LINEAR), /255, CHW."""
    out = cv_resize_linear_u8(img, dim[0], dim[1])
    return torch.from_numpy((out.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1).copy())


def correct_yolo_boxes(bboxes, org_w, org_h, img_w, img_h, is_letterbox=False):
    """reference boundingbox.py:95-149: un-letterbox (or un-resize) x1y1x2y2 boxes, clip, convert to xywh."""
    if len(bboxes) == 0:
        return bboxes
    b = bboxes.clone().float()
    mask = b.sum(-1) != 0
    if is_letterbox:
        ratio = min(img_w / org_w, img_h / org_h)
        rw, rh = int(org_w * ratio), int(org_h * ratio)
        xp, yp = (img_w - rw) // 2, (img_h - rh) // 2
        b[mask, 0] = torch.clamp((b[mask, 0] - xp) / ratio, 0, org_w)
        b[mask, 2] = torch.clamp((b[mask, 2] - xp) / ratio, 0, org_w)
        b[mask, 1] = torch.clamp((b[mask, 1] - yp) / ratio, 0, org_h)
        b[mask, 3] = torch.clamp((b[mask, 3] - yp) / ratio, 0, org_h)
    else:
        rx, ry = img_w / org_w, img_h / org_h
        b[mask, 0] = torch.clamp(b[mask, 0] / rx, 0, org_w)
        b[mask, 2] = torch.clamp(b[mask, 2] / rx, 0, org_w)
        b[mask, 1] = torch.clamp(b[mask, 1] / ry, 0, org_h)
        b[mask, 3] = torch.clamp(b[mask, 3] / ry, 0, org_h)
    return torch.stack((b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]), -1)
