"""Cycle split of one workgroup of the four-wave 192x128 kernel (conv_planes_w4.hip, tile code 12) -- needs a -DYV3_TIMELINE build
(tools/build_variant.sh tl "-DYV3_TIMELINE"; YV3_MEASURE=1 YV3_LIB=.../libyv3_tl.so).  Per wave: prologue | per chunk: k-step 0
(18 MFMAs + k-step 1's fragment reads), wait (vmcnt / lgkmcnt before the barrier), barrier, k-step 1 (18 MFMAs + DMA pieces + next chunk's
reads) | epilogue.  WG = workgroup index sampled (default 700: a later round, warm instruction cache; 100 = first round)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolo_v3_amd import _ffi, engine
from yolo_v3_amd.darknet import conv_bn_relu
torch.cuda.set_device(0)
lib = _ffi.lib()
B = int(os.environ.get("BB", "64"))
WG = int(os.environ.get("WG", "700"))
LAYERS = {"c52": (128, 256, 3, 52), "c26": (256, 512, 3, 26), "c104": (64, 128, 3, 104), "L52": (512, 256, 3, 52), "p26": (512, 256, 1, 26)}
for name in (sys.argv[1:] or ["c52", "c26", "c104", "L52"]):
    cin, cout, k, H = LAYERS[name]
    m = conv_bn_relu(cin, cout, k, 1).cuda().eval()
    pc = engine.pack_conv(m, m._spec(), _ffi.F32H2)
    x = engine.to_planes(torch.rand(B, H, H, cin, device="cuda") - 0.5, _ffi.F32H2)
    r = engine.to_planes(torch.rand(B, H, H, cout, device="cuda") - 0.5, _ffi.F32H2) if k == 3 else None
    y = engine.alloc_act(B, H, H, cout, _ffi.F32H2, "cuda")
    d = engine.make_desc(pc, x, y, B, H, H, r, dtype=_ffi.F32H2)
    d.options = (d.options & ~(0xff << 8)) | (12 << 8)
    d.tune[2] = WG
    for _ in range(3):
        _ffi.check(lib.yv3_conv2d(d, _ffi.stream_ptr()))
    torch.cuda.synchronize()
    a = pc.alpha.cpu()[:32].view(4, 8)
    a2 = pc.alpha.cpu()[32:40].view(4, 2)
    print(name, "B =", B, "workgroup", WG)
    for w in range(4):
        pro, k0, wait, bar, k1, epi, nk, tot = a[w].tolist()
        idx, iss = a2[w].tolist()
        pro += idx + iss
        print("  wave %d: prologue %6.0f (index math %5.0f, DMA issue of 2 chunks %5.0f, wait for chunk 0 + its fragments %5.0f) | per chunk: k-step0 %5.0f  wait %5.0f  barrier %5.0f  k-step1 %5.0f (sum %5.0f; 1152 MFMA cycles) | epilogue %6.0f | chunks %d total %7.0f cycles"
              % (w, pro, idx, iss, pro - idx - iss, k0, wait, bar, k1, k0 + wait + bar + k1, epi, nk, tot))
