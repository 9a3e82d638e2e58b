"""Soak of the round-3 paths: many Detector calls per (mode, batch) with the per-launch Winograd choice active; every call must
reproduce the first call's boxes bit for bit (same input, same schedule) and leave the kernels' status word clean.  With
YV3_WINO_EVEN=1 YV3_WINO_ALWAYS=1 in the environment the Winograd stage runs its stream-K schedule on every eligible layer
(hand-over flags must end every launch reset)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import synth, Detector, _ffi
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
n = int(os.environ.get("N", "150"))
net = load_sw1_net(synth.weight_stream()).cuda()
for mode, name in ((_ffi.F32H2, "f32h2"), (_ffi.F32, "f32"), (_ffi.BF16, "bf16")):
    for B in (64, 32, 16, 1):
        x = torch.from_numpy(synth.images(B, 416, 3 + B)).cuda()
        d = Detector(net, B, 416, 416, dtype=mode)
        first = d(x)
        nw = sum(1 for p in d.lane_plans for i in range(p.n_desc) if p.descs[i].w_wino)
        t0 = time.perf_counter()
        for i in range(n if mode != _ffi.F32 else n // 3):
            r = d(x)
            assert len(r) == len(first) and all(torch.equal(a, b) for a, b in zip(first, r)), "results changed between calls (%s B=%d call %d)" % (name, B, i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (n if mode != _ffi.F32 else n // 3)
        assert int(d.plan.flags.item()) == 0
        for p in d.lane_plans:
            if p.wino_ws is not None:                   # hand-over flags of the even schedule: last 512 ints (+pad) of the workspace
                assert int(p.wino_ws[-(512 * 4 + 256):].view(torch.int32).abs().sum()) == 0, "a hand-over flag was left set"
        print("%-5s B=%-2d lanes=%d descriptors with Winograd filters=%d : %d calls identical, %.3f ms/call" % (name, B, d.lanes, nw, n, dt * 1e3)); sys.stdout.flush()
