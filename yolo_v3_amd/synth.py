"""Deterministic synthetic weights and inputs ("SW-1" / "SW-dense").

``yolov3.weights`` is not available (reference ``weights/.gitignore``; no network), so
benchmarks and parity tests use weights generated from an integer counter RNG
(splitmix64) with float32-only arithmetic: the stream is bit-identical on any
machine without shipping a file.  The stream is laid out exactly like a darknet
``.weights`` file (reference ``darknet.py:265-290``): per conv_bn_relu
``bn.bias, bn.weight, running_mean, running_var, conv.weight``; per plain conv
``bias, weight``; convs in cfg order (``arch.conv_specs``).
"""
import numpy as np

from . import arch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, stream, n):
    """n float32 values in [0,1) (24-bit mantissa, exact), keyed by (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
                           + np.uint64(stream) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(1))
        idx = np.arange(n, dtype=np.uint64)
        v = _splitmix64(base + idx)
    return ((v >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def uniform(seed, stream, n, lo, hi):
    u = uniform01(seed, stream, n)
    lo = np.float32(lo)
    hi = np.float32(hi)
    return lo + u * (hi - lo)          # float32 mul/add only -> reproducible everywhere


WEIGHT_SEED = 20260928


def weight_stream(num_class=80, seed=WEIGHT_SEED, b_obj=-2.25, b_cls=0.0, head_gain=1.0):
    """float32 array in darknet stream order for the full YoloNet (62 001 757 values at 80 classes)."""
    specs = arch.conv_specs(num_class)
    out = np.empty(arch.floats_in_stream(specs), dtype=np.float32)
    p = 0
    attrib = num_class + 5
    for i, sp in enumerate(specs):
        fan_in = sp.cin * sp.k * sp.k
        nw = sp.cout * fan_in
        if sp.bn:
            gscale = np.float32(0.35) if sp.res2 else np.float32(1.0)
            out[p:p + sp.cout] = uniform(seed, i * 8 + 0, sp.cout, -0.17, 0.17); p += sp.cout      # bn.bias
            out[p:p + sp.cout] = uniform(seed, i * 8 + 1, sp.cout, 0.6, 1.2) * gscale; p += sp.cout  # bn.weight
            out[p:p + sp.cout] = uniform(seed, i * 8 + 2, sp.cout, -0.17, 0.17); p += sp.cout      # running_mean
            out[p:p + sp.cout] = uniform(seed, i * 8 + 3, sp.cout, 0.7, 1.4); p += sp.cout         # running_var
            a = np.float32(np.sqrt(6.0 / (1.01 * fan_in)))
            out[p:p + nw] = uniform(seed, i * 8 + 4, nw, -a, a); p += nw
        else:
            bias = np.zeros(sp.cout, dtype=np.float32)
            for anchor in range(3):
                bias[anchor * attrib + 4] = b_obj
                bias[anchor * attrib + 5:(anchor + 1) * attrib] = b_cls
            out[p:p + sp.cout] = bias; p += sp.cout
            a = np.float32(head_gain * np.sqrt(3.0 / fan_in))
            out[p:p + nw] = uniform(seed, i * 8 + 4, nw, -a, a); p += nw
    assert p == out.size
    return out


def dense_weight_stream(num_class=80, seed=WEIGHT_SEED):
    """"SW-dense": same weights, head biases raised so thousands of rows pass conf 0.5."""
    return weight_stream(num_class, seed, b_obj=1.0, b_cls=1.0)


def eval_weight_stream(num_class=80, seed=WEIGHT_SEED):
    """"SW-eval": same weights, head biases lowered (objectness -7, classes -3) so that at the reference's
    evaluation thresholds (conf 0.005 / nms 0.45, is_eval=True: evaluate.py:201-204) an image yields ~1-2 k
    (row, class) candidates, a few hundred in the busiest class -- the regime of a trained network, instead of the
    > 5e5 pairs SW-1's zero class biases produce at 0.005."""
    return weight_stream(num_class, seed, b_obj=-7.0, b_cls=-3.0)


def write_darknet_weights(path, stream, seen=0):
    """Write ``stream`` as a darknet file: int32 {major=0, minor=2, revision=0}, int64 seen, floats."""
    with open(path, "wb") as fp:
        np.array([0, 2, 0], dtype=np.int32).tofile(fp)
        np.array([seen], dtype=np.int64).tofile(fp)
        np.ascontiguousarray(stream, dtype=np.float32).tofile(fp)


def stream_to_state_dict(stream, num_class=80):
    """Split a darknet stream into {state_dict key: ndarray} (reference key names, OIHW weights)."""
    sd = {}
    p = 0
    for sp in arch.conv_specs(num_class):
        nw = sp.cout * sp.cin * sp.k * sp.k
        shape = (sp.cout, sp.cin, sp.k, sp.k)
        if sp.bn:
            for key in ("bn.bias", "bn.weight", "bn.running_mean", "bn.running_var"):
                sd["%s.%s" % (sp.name, key)] = stream[p:p + sp.cout]; p += sp.cout
            sd["%s.conv.weight" % sp.name] = stream[p:p + nw].reshape(shape); p += nw
        else:
            sd["%s.bias" % sp.name] = stream[p:p + sp.cout]; p += sp.cout
            sd["%s.weight" % sp.name] = stream[p:p + nw].reshape(shape); p += nw
    assert p == stream.size
    return sd


def noise_images(batch, size, seed):
    """[batch,3,size,size] float32 NCHW, i.i.d. U[0,1)."""
    n = batch * 3 * size * size
    return uniform01(seed, 0x494D47, n).reshape(batch, 3, size, size)


def images(batch, size, seed, n_rect=24):
    """[batch,3,size,size] float32 NCHW in [0,1]: synthetic scenes.

    Flat background + ``n_rect`` axis-aligned rectangles of random colour per image + mild
    pixel noise.  Pure i.i.d. noise averages out in the deep layers (every image then yields
    the same detections); rectangles keep spatial structure so different images / positions
    produce different candidate sets.  Integer RNG + float32 mul/add only: bit-reproducible.
    """
    out = np.empty((batch, 3, size, size), dtype=np.float32)
    for b in range(batch):
        u = uniform01(seed, 0x5343454E45 + b, 3 + n_rect * 7)
        img = np.empty((3, size, size), dtype=np.float32)
        img[:] = u[:3].reshape(3, 1, 1)
        for r in range(n_rect):
            q = u[3 + r * 7: 3 + (r + 1) * 7]
            x0 = int(q[0] * np.float32(size)); y0 = int(q[1] * np.float32(size))
            w = 8 + int(q[2] * np.float32(size * 0.45)); h = 8 + int(q[3] * np.float32(size * 0.45))
            img[:, y0:min(size, y0 + h), x0:min(size, x0 + w)] = q[4:7].reshape(3, 1, 1)
        noise = uniform01(seed, 0x4E4F49 + b * 16, 3 * size * size).reshape(3, size, size)
        img = img + (noise - np.float32(0.5)) * np.float32(0.12)
        out[b] = np.minimum(np.maximum(img, np.float32(0.0)), np.float32(1.0))
    return out
