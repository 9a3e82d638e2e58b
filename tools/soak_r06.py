"""Soak of the round-6 exact-fp32 plans: the F(4x4,3x3) stage's even schedule (items of small launches / small tails cut into ranges of patch
rows, summed through the hand-over area) and the persistent 1x1 GEMM + rest-on-tiles split -- many Detector calls per (batch, size, graph):
every call bit for bit the first one, the status word clean, every hand-over flag back to zero.   N=150 python tools/soak_r06.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import synth, Detector, _ffi
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
n = int(os.environ.get("N", "150"))
stream = synth.weight_stream()
SK = 512 * (512 * 128 * 4 + 4) + 256                     # yv3_wino_sk_bytes(): the hand-over area is the buffer's tail, its flags sit in the last 128 KB part
for B, size, graph in ((1, 416, False), (1, 416, True), (2, 416, False), (4, 416, False), (8, 416, True), (12, 416, False), (24, 416, False),
                       (42, 416, False), (64, 416, False), (1, 608, False), (8, 608, False), (16, 608, False)):
    net = load_sw1_net(stream).cuda()
    x = torch.from_numpy(synth.images(min(B, 16), size, 3)).cuda().repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
    d = Detector(net, B, size, size, dtype=_ffi.F32, graph=graph)
    first = d(x)
    t0 = time.perf_counter()
    for i in range(n):
        r = d(x)
        assert len(r) == len(first) and all(torch.equal(a, b) for a, b in zip(first, r)), "results changed between calls"
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    plans = d.lane_plans if d.lanes > 1 else [d.plan]
    cut = gemm2 = 0
    for p in plans:
        assert int(p.flags.item()) == 0
        ws = p.wino_ws
        off = (ws.numel() - SK) & ~255
        flags = ws[off + 1023 * 128 * 1024: off + 1023 * 128 * 1024 + 4 * 1023].view(torch.int32)
        assert int(flags.abs().sum()) == 0, "a hand-over flag was left set"
        cut += int(ws[off: off + 1023 * 128 * 1024].view(torch.int32).abs().sum() != 0)          # the area was used
        gemm2 += sum(1 for (_, f), k in zip(p.forms(), p.launches()) if f == 0 and k == 2)
    print("exact fp32 B=%d %dx%d graph=%d lanes=%d: %d calls identical, %.3f ms/call (%d boxes); F(4x4) launches %d, hand-over area used: %s, 1x1 layers as GEMM + tiles: %d"
          % (B, size, size, graph, d.lanes, n, dt * 1e3, sum(int(b.shape[0]) for b in first if b.numel()),
             sum(f == 2 for p in plans for _, f in p.forms()), bool(cut), gemm2))
    sys.stdout.flush()
