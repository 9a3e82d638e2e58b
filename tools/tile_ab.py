"""Same-box A/B of conv tile variants (yv3_conv_desc.options tile code) on single layers through the C-ABI: interleaved timing
+ bitwise comparison of the outputs (same K order => the variants must agree bit for bit).
  DT=bf16 BB=16 python tools/tile_ab.py 0,5 c76 c38 c19      # variants = YV3_TILE codes (0 = the shipped selection)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu

LAYERS = {  # name: cin, cout, k, stride, H (input), res
    "c52": (128, 256, 3, 1, 52, True), "c26": (256, 512, 3, 1, 26, True), "c13": (512, 1024, 3, 1, 13, True), "c104": (64, 128, 3, 1, 104, True),
    "c76": (128, 256, 3, 1, 76, True), "c38": (256, 512, 3, 1, 38, True), "c19": (512, 1024, 3, 1, 19, True), "c152": (64, 128, 3, 1, 152, True),
    "d76": (128, 256, 3, 2, 152, False), "d38": (256, 512, 3, 2, 76, False),
    "p76": (256, 128, 1, 1, 76, False), "p38": (512, 256, 1, 1, 38, False), "p19": (1024, 512, 1, 1, 19, False),
    "p52": (256, 128, 1, 1, 52, False), "p26": (512, 256, 1, 1, 26, False), "p13": (1024, 512, 1, 1, 13, False),
    "L52": (512, 256, 3, 1, 52, True), "s104": (64, 128, 3, 2, 208, False), "s52": (128, 256, 3, 2, 104, False), "s26": (256, 512, 3, 2, 52, False), "s13": (512, 1024, 3, 2, 26, False),
}
B = int(os.environ.get("BB", "16"))
iters = int(os.environ.get("ITERS", "20"))
dt = {"f32": _ffi.F32, "bf16": _ffi.BF16, "f32x3": _ffi.F32X3, "f32h2": _ffi.F32H2}[os.environ.get("DT", "bf16")]
variants = [int(v) for v in sys.argv[1].split(",")]
names = sys.argv[2:] or ["c76", "c38", "c19"]
torch.cuda.set_device(0)
lib = _ffi.lib()
st = _ffi.stream_ptr()
for name in names:
    cin, cout, k, s, H, res = LAYERS[name]
    m = conv_bn_relu(cin, cout, k, s).cuda().eval()
    pc = engine.pack_conv(m, m._spec(), dt)
    ho, wo = engine.out_hw(H, H, k, s)
    zero = os.environ.get("ZERO") == "1"            # all-zero operands: the matrix pipes draw far less power, the chip is no longer power-limited
    if zero:
        with torch.no_grad():
            m.conv.weight.zero_()
        pc = engine.pack_conv(m, m._spec(), dt)
    hib = int(os.environ.get("HIBITS", "0"))         # energy probe: operands whose fp16 HI part carries only `hib` significant bits (low bits 0),
                                                     # the LO part a random 11-bit residue below half an ulp of fp16: hi + lo as the kernels see them
    def sparse_hi(t):
        if not hib:
            return t
        m, e = torch.frexp(t.float())
        q = torch.ldexp(torch.round(m * (1 << hib)) / (1 << hib), e)                 # `hib` significant bits
        ulp16 = torch.ldexp(torch.ones_like(q), e - 11)                              # fp16 ulp at this magnitude
        return q + (torch.rand_like(q) - 0.5) * 0.98 * ulp16                         # |residue| < half an fp16 ulp: RN16(value) == q
    if hib:
        with torch.no_grad():
            m.conv.weight.copy_(sparse_hi(m.conv.weight))
        pc = engine.pack_conv(m, m._spec(), dt)
    x = engine.to_planes(sparse_hi(torch.rand(B, H, H, cin, device="cuda") - 0.5) * (0.0 if zero else 1.0), dt)
    r = engine.to_planes((torch.rand(B, ho, wo, cout, device="cuda") - 0.5) * (0.0 if zero else 1.0), dt) if res else None
    fl = 2.0 * B * ho * wo * cout * cin * k * k
    descs, outs = [], []
    for v in variants:
        y = engine.alloc_act(B, ho, wo, cout, dt, "cuda")
        y.zero_()
        d = engine.make_desc(pc, x, y, B, H, H, r, dtype=dt)
        if v >= 1000:
            d.tune[1] = v - 1000                                          # codes 1000 + t: the shipped tile selection with tune[1] = t (conv_planes.hip)
        else:
            d.options = (d.options & ~(0xff << 8)) | ((v % 100) << 8)
        if 200 <= v < 1000:
            d.options |= _ffi.OPT_K3S1                                    # codes 2xx: the kw-tap-reuse kernel (conv_planes_k3s1.hip)
        descs.append(d); outs.append(y)
        for _ in range(3):
            _ffi.check(lib.yv3_conv2d(d, st))
    torch.cuda.synchronize()
    same = [bool(torch.equal(outs[0], o)) for o in outs]
    f0 = engine.from_planes(outs[0], dt).float()
    dmax = [float((engine.from_planes(o, dt).float() - f0).abs().max()) for o in outs]
    best = [1e9] * len(variants)
    for rep in range(3):                         # interleaved passes
        for i, d in enumerate(descs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                _ffi.check(lib.yv3_conv2d(d, st))
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters)
    print("%-5s B=%d %dx%d %d->%d k%d s%d :" % (name, B, H, H, cin, cout, k, s) +
          "".join("  tile %d: %.4f ms %.0f TF%s" % (v, t, fl / t / 1e9, "" if ok else " (differs: max|d| %.3g)" % dm) for v, t, ok, dm in zip(variants, best, same, dmax)))
    sys.stdout.flush()
