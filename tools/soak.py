"""Soak: many Detector calls in both schedules; checks the status word stays clean and results stay identical
call to call (same input -> same boxes, bitwise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import synth, Detector, _ffi
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
n = int(os.environ.get("N", "300"))
for sk in (False, True):
    net = load_sw1_net(synth.weight_stream()).cuda()
    net.stream_k = sk
    for B in (64, 5, 1):
        x = torch.from_numpy(synth.images(min(B, 16), 416, 3)).cuda().repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
        d = Detector(net, B, 416, 416)
        first = d(x)
        t0 = time.perf_counter()
        for i in range(n):
            r = d(x)
            if i % 50 == 0:
                assert all(torch.equal(a, b) for a, b in zip(first, r)), "results changed between calls"
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        assert int(d.plan.flags.item()) == 0
        if d.plan.workspace is not None:
            assert int(d.plan.workspace[-4 * 512:].view(torch.int32).abs().sum()) == 0
        print("stream_k=%d B=%d: %d calls ok, %.3f ms/call (%.0f img/s incl. host sync + result lists)" % (sk, B, n, dt * 1e3, B / dt)); sys.stdout.flush()
