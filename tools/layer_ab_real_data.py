"""Tile-code A/B of single layers ON THE NETWORK'S OWN DATA: runs one forward of the bench workload, then re-launches chosen descriptors of the
plan (their real input / residual / weight buffers) with forced tile codes, interleaved, and checks the outputs bit for bit.  Answers "does an
isolated A/B on uniform random operands carry over to real activations?"  (round 5: the bf16 ping-pong tiles did not).
   DT=bf16 SIZE=608 BB=16 python tools/layer_ab_real_data.py 11,15 256,512,3 128,256,3      (codes; then cin,cout,k of the layers to test)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolo_v3_amd import _ffi, arch, synth
from yolo_v3_amd._ffi import ConvDesc
B, size = int(os.environ.get("BB", "16")), int(os.environ.get("SIZE", "608"))
mode = {"bf16": _ffi.BF16, "f32h2": _ffi.F32H2}[os.environ.get("DT", "bf16")]
iters = int(os.environ.get("ITERS", "20"))
codes = [int(c) for c in sys.argv[1].split(",")]
want = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
net = bench.make_net(synth.weight_stream(), size, dev)
net.math_mode = mode
x = bench.scenes(B, size, 2, dev)
with torch.no_grad():
    net.forward_cat(x)
torch.cuda.synchronize()
plan = net.engine().plan(B, size, size)
specs = arch.conv_specs(); hw = arch.conv_output_hw(size)
lib = _ffi.lib(); st = _ffi.stream_ptr()
seen = set()
for j in range(plan.first_desc, plan.n_desc):
    si = plan.desc_spec[j]; sp = specs[si]
    key = (sp.cin, sp.cout, sp.k)
    if key not in want or (key, hw[si][0], sp.stride) in seen or sp.stride != 1:
        continue
    seen.add((key, hw[si][0], sp.stride))
    src = plan.descs[j]
    ybytes = 2 * B * hw[si][0] * hw[si][1] * sp.cout * (2 if mode == _ffi.F32H2 else 1)
    descs, outs = [], []
    for c in codes:
        d = ConvDesc(); ctypes.memmove(ctypes.byref(d), ctypes.byref(src), ctypes.sizeof(d))
        y = torch.zeros(ybytes, dtype=torch.uint8, device=dev)
        d.y = y.data_ptr()
        d.options = (d.options & ~(0xff << 8)) | (c << 8)
        descs.append(d); outs.append(y)
        for _ in range(3):
            _ffi.check(lib.yv3_conv2d(ctypes.byref(d), st))
    torch.cuda.synchronize()
    same = [bool(torch.equal(o, outs[0])) for o in outs]
    best = [1e9] * len(codes)
    for rep in range(3):
        for i, d in enumerate(descs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                _ffi.check(lib.yv3_conv2d(ctypes.byref(d), st))
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters)
    fl = 2.0 * B * hw[si][0] * hw[si][1] * sp.cout * sp.cin * sp.k * sp.k
    print("%s @%d B=%d (network data, layer %d):" % (key, hw[si][0], B, si) +
          "".join("  tile %d: %.4f ms %.0f TF%s" % (c, t, fl / t / 1e9, "" if ok else " (DIFFERS)") for c, t, ok in zip(codes, best, same)))
    sys.stdout.flush()
