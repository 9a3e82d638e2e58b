"""Static description of the YOLOv3 / Darknet-53 convolution stack.

One table drives everything that needs to agree on layer order and shapes:
the darknet ``.weights`` stream order (reference ``darknet.py:292-303`` walks the
module tree depth-first, which equals the darknet cfg order), the synthetic
weight generator, the parameter containers in ``darknet.py`` and the HIP
execution plan in ``engine.py``.

Reference architecture: ``darknet.py:72-81`` (backbone ``Darknet([1,2,8,8,4])``),
``darknet.py:107-120`` (``PreDetectionConvGroup``), ``darknet.py:153-162``
(``UpsampleGroup``), wiring ``darknet.py:179-194``.
"""
from collections import namedtuple

# name  : state_dict prefix of the module that owns the convolution
# bn    : True  -> conv(no bias) + BatchNorm(eval) + LeakyReLU(0.1)   (reference conv_bn_relu)
#         False -> plain conv with bias, no activation                 (the three head convs)
# res2  : True for the second conv of a res_layer (used by the synthetic weight recipe)
ConvSpec = namedtuple("ConvSpec", "name cin cout k stride bn res2")

BACKBONE_BLOCKS = (1, 2, 8, 8, 4)
DEFAULT_ANCHORS = (10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326)
ANCHOR_MASKS = ((6, 7, 8), (3, 4, 5), (0, 1, 2))


def backbone_specs(blocks=BACKBONE_BLOCKS, nout=32):
    """52 convolutions of the Darknet-53 feature extractor, cfg order."""
    specs = [ConvSpec("feature.mlist.0", 3, nout, 3, 1, True, False)]
    idx = 1
    for stage, nblk in enumerate(blocks):
        cin = nout * (2 ** stage)
        specs.append(ConvSpec("feature.mlist.%d" % idx, cin, cin * 2, 3, 2, True, False))
        idx += 1
        for _ in range(nblk):
            c = cin * 2
            specs.append(ConvSpec("feature.mlist.%d.conv1" % idx, c, c // 2, 1, 1, True, False))
            specs.append(ConvSpec("feature.mlist.%d.conv2" % idx, c // 2, c, 3, 1, True, True))
            idx += 1
    return specs


def predet_specs(prefix, nin, nout, num_class):
    """7 convolutions of one detection branch: (1x1 n, 3x3 2n) x3 then plain 1x1 -> 3*(5+C)."""
    specs = []
    cin = nin
    for i in range(3):
        specs.append(ConvSpec("%s.mlist.%d" % (prefix, 2 * i), cin, nout, 1, 1, True, False))
        specs.append(ConvSpec("%s.mlist.%d" % (prefix, 2 * i + 1), nout, nout * 2, 3, 1, True, False))
        cin = nout * 2
    specs.append(ConvSpec("%s.mlist.6" % prefix, cin, (num_class + 5) * 3, 1, 1, False, False))
    return specs


def conv_specs(num_class=80):
    """All 75 convolutions in darknet ``.weights`` stream order."""
    s = backbone_specs()
    s += predet_specs("pre_det1", 1024, 512, num_class)
    s.append(ConvSpec("up1.conv", 512, 256, 1, 1, True, False))
    s += predet_specs("pre_det2", 768, 256, num_class)
    s.append(ConvSpec("up2.conv", 256, 128, 1, 1, True, False))
    s += predet_specs("pre_det3", 384, 128, num_class)
    return s


def floats_in_stream(specs):
    """Number of float32 values a darknet .weights stream holds for ``specs``."""
    n = 0
    for sp in specs:
        n += sp.cout * sp.cin * sp.k * sp.k
        n += 4 * sp.cout if sp.bn else sp.cout
    return n


def conv_macs_per_image(size, num_class=80):
    """Multiply-accumulates of the 75 convolutions for one ``size`` x ``size`` image."""
    total = 0
    for sp, (h, w) in zip(conv_specs(num_class), conv_output_hw(size, num_class)):
        total += h * w * sp.cout * sp.cin * sp.k * sp.k
    return total


def conv_output_hw(size, num_class=80):
    """Output (H, W) of every conv in ``conv_specs`` order for a square input."""
    out = []
    specs = conv_specs(num_class)
    s = size
    # backbone
    i = 0
    for sp in specs[:52]:
        if sp.stride == 2:
            s = (s + 2 * 1 - 3) // 2 + 1
        out.append((s, s))
        i += 1
    g1 = s
    out += [(g1, g1)] * 7          # pre_det1
    out.append((g1, g1))           # up1.conv (before the x2 upsample)
    g2 = g1 * 2
    out += [(g2, g2)] * 7          # pre_det2
    out.append((g2, g2))           # up2.conv
    g3 = g2 * 2
    out += [(g3, g3)] * 7          # pre_det3
    return out
