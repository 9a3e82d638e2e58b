"""Lane counts / splits on stream sets known to run concurrently (consecutively created torch streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import engine, synth
from yolo_v3_amd.darknet import YoloNet, WeightManager
B, S = int(os.environ.get("BB", "64")), int(os.environ.get("SIZE", "416"))
torch.cuda.set_device(0)
net = YoloNet((S, S)).eval(); WeightManager(net).load_stream(synth.weight_stream()); net = net.cuda()
x = torch.from_numpy(synth.images(16, S, 7)).cuda().repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
eng = net.engine(); eng.ensure_packed()
pool = [torch.cuda.Stream() for _ in range(4)]
def t(fn, iters=15):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
def case(sizes):
    plans = [engine.Plan(eng, n, S, S) for n in sizes]
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    dets = torch.empty((B, plans[0].N, plans[0].attrib), device="cuda")
    def step():
        main = torch.cuda.current_stream(); ev = torch.cuda.Event(); ev.record(main)
        for i, p in enumerate(plans):
            st = pool[i] if len(plans) > 1 else main
            if len(plans) > 1: st.wait_event(ev)
            with torch.cuda.stream(st):
                eng.run_convs(p, x[offs[i]:offs[i] + sizes[i]], dets[offs[i]:offs[i] + sizes[i]])
            if len(plans) > 1:
                e = torch.cuda.Event(); e.record(st); main.wait_event(e)
    return t(step)
for sizes in ([B], [B // 2, B - B // 2], [9 * B // 16, B - 9 * B // 16], [5 * B // 8, B - 5 * B // 8], [B // 3, B // 3, B - 2 * (B // 3)],
              [B // 2, B // 4, B - B // 2 - B // 4], [B // 4] * 4, [B], [B // 2, B - B // 2]):
    ms = case(sizes)
    print("B=%d %dx%d lanes %-16s: %.3f ms -> %.0f img/s (convs only)" % (B, S, S, sizes, ms, B / ms * 1e3)); sys.stdout.flush()
