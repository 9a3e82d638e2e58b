"""Does running the batch as L sub-batches on L HIP streams (concurrent kernels fill each other's tails and
overlap memory-bound phases with MFMA-bound ones) beat one launch sequence over the whole batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine, synth
from yolo_v3_amd.darknet import YoloNet, WeightManager

B = int(os.environ.get("BB", "64")); S = int(os.environ.get("SIZE", "416")); iters = int(os.environ.get("ITERS", "10"))
torch.cuda.set_device(0)
net = YoloNet((S, S)).eval()
stream = synth.weight_stream()
assert WeightManager(net).load_stream(stream) == stream.size
net = net.cuda()

x = torch.from_numpy(synth.images(B, S, 7)).cuda()
eng = net.engine()
eng.ensure_packed()
for L in (1, 2, 4):
    sub = B // L
    plans = [engine.Plan(eng, sub, S, S) for _ in range(L)]
    streams = [torch.cuda.Stream() for _ in range(L)]
    dets = torch.empty((B, plans[0].N, plans[0].attrib), device="cuda")
    xs = [x[i * sub:(i + 1) * sub] for i in range(L)]
    def step():
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        for i, st in enumerate(streams):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                eng.run_convs(plans[i], xs[i], dets[i * sub:(i + 1) * sub])
                eng.run_decode(plans[i], dets[i * sub:(i + 1) * sub])
            e = torch.cuda.Event(); e.record(st); main.wait_event(e)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print("lanes=%d (sub-batch %d): %.3f ms per %d images -> %.0f img/s (convs + decode only)" % (L, sub, ms, B, B / ms * 1e3)); sys.stdout.flush()
