"""Two sub-batches on two HIP streams: is the speed-up a property of the stream PAIR (hardware-queue mapping) or of the run?"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine, synth
from yolo_v3_amd.darknet import YoloNet, WeightManager
B, S = 64, 416
torch.cuda.set_device(0)
net = YoloNet((S, S)).eval(); WeightManager(net).load_stream(synth.weight_stream()); net = net.cuda()
x = torch.from_numpy(synth.images(16, S, 7)).cuda().repeat(4, 1, 1, 1).contiguous()
eng = net.engine(); eng.ensure_packed()
plans = [engine.Plan(eng, 32, S, S) for _ in range(2)]
one = engine.Plan(eng, 64, S, S)
dets = torch.empty((B, one.N, one.attrib), device="cuda")
def t(fn, iters=15):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
def single():
    eng.run_convs(one, x, dets)
print("single stream bs=64: %.3f ms" % t(single))
pool = [torch.cuda.Stream() for _ in range(8)] + [torch.cuda.Stream(priority=-1) for _ in range(4)]
def lanes(sa, sb):
    def step():
        main = torch.cuda.current_stream(); ev = torch.cuda.Event(); ev.record(main)
        for i, st in enumerate((sa, sb)):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                eng.run_convs(plans[i], x[32 * i:32 * i + 32], dets[32 * i:32 * i + 32])
            e = torch.cuda.Event(); e.record(st); main.wait_event(e)
    return step
for (a, b) in [(0, 1), (0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 5), (2, 6), (0, 8), (0, 9), (8, 9), (3, 7), (0, 1)]:
    print("streams (%d,%d) [%x,%x]: %.3f ms" % (a, b, pool[a].cuda_stream, pool[b].cuda_stream, t(lanes(pool[a], pool[b]))))
print("single stream bs=64: %.3f ms" % t(single))
