"""Soak of the round-5 default plans (four-wave w4 tile at bs 16-64 one lane, 1x1 layers under two lanes; exact-fp32 four-wave Winograd tile): many Detector calls per (mode, batch, size) -- every call's boxes must equal the first call's
bit for bit, the status word must stay clean and the stream-K flags must be back to zero.  Covers the automatic stream-K range
(<= 1536 cells), the two-lane Winograd plan, the bf16 192-row / 256x256 tiles and HIP-graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import synth, Detector, _ffi
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
n = int(os.environ.get("N", "150"))
stream = synth.weight_stream()
for mode, B, size, graph in ((_ffi.F32H2, 1, 416, True), (_ffi.F32H2, 2, 416, False), (_ffi.F32H2, 4, 608, False), (_ffi.F32H2, 9, 416, False),
                             (_ffi.F32H2, 9, 416, True), (_ffi.F32H2, 10, 416, False), (_ffi.F32H2, 64, 416, False), (_ffi.BF16, 16, 608, False),
                             (_ffi.BF16, 8, 608, False), (_ffi.BF16, 64, 416, False), (_ffi.F32, 16, 416, False), (_ffi.F32, 64, 416, False), (_ffi.F32, 16, 608, False), (_ffi.F32H2, 16, 416, False), (_ffi.F32H2, 32, 416, False), (_ffi.F32H2, 32, 416, True), (_ffi.F32H2, 16, 608, False), (_ffi.F32H2, 8, 608, False), (_ffi.F32, 32, 416, False), (_ffi.F32, 64, 416, False), (_ffi.F32H2, 128, 416, False)):
    net = load_sw1_net(stream).cuda()
    x = torch.from_numpy(synth.images(min(B, 16), size, 3)).cuda().repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
    d = Detector(net, B, size, size, dtype=mode, graph=graph)
    first = d(x)
    t0 = time.perf_counter()
    for i in range(n):
        r = d(x)
        if True:
            assert len(r) == len(first) and all(torch.equal(a, b) for a, b in zip(first, r)), "results changed between calls"
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    plans = d.lane_plans if d.lanes > 1 else [d.plan]
    for p in plans:
        assert int(p.flags.item()) == 0
        if p.workspace is not None:
            assert int(p.workspace[-4 * 512:].view(torch.int32).abs().sum()) == 0
    print("mode %d B=%d %dx%d graph=%d lanes=%d stream-K=%s: %d calls identical, %.3f ms/call (%d boxes)"
          % (mode, B, size, size, graph, d.lanes, plans[0].workspace is not None, n, dt * 1e3, sum(int(b.shape[0]) for b in first if b.numel())))
    sys.stdout.flush()
