"""Does running the batch as L sub-batches on L HIP streams (concurrent kernels fill each other's partial rounds and overlap
memory-bound phases with MFMA-bound ones) beat one launch sequence over the whole batch?  Variants: even / uneven splits,
and lane 1 started half a network behind lane 0 (so that unlike layers overlap)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import _ffi, engine, synth
from yolo_v3_amd.darknet import YoloNet, WeightManager
from yolo_v3_amd._ffi import ConvDesc

B = int(os.environ.get("BB", "64")); S = int(os.environ.get("SIZE", "416")); iters = int(os.environ.get("ITERS", "20"))
torch.cuda.set_device(0)
net = YoloNet((S, S)).eval()
stream = synth.weight_stream()
assert WeightManager(net).load_stream(stream) == stream.size
net = net.cuda()
x = torch.from_numpy(synth.images(min(B, 16), S, 7)).cuda().repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
eng = net.engine()
eng.ensure_packed()
lib = _ffi.lib()

def seq(plan, dets, lo, hi):
    plan.bind_detections(dets)
    tail = ctypes.cast(ctypes.addressof(plan.descs) + lo * ctypes.sizeof(ConvDesc), ctypes.POINTER(ConvDesc))
    _ffi.check(lib.yv3_conv2d_sequence(tail, hi - lo, _ffi.stream_ptr()))

def run(splits, stagger):
    plans = [engine.Plan(eng, n, S, S) for n in splits]
    streams = [torch.cuda.Stream() for _ in splits]
    offs = [sum(splits[:i]) for i in range(len(splits))]
    dets = torch.empty((B, plans[0].N, plans[0].attrib), device="cuda")
    def step():
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        mid_ev = None
        for i, st in enumerate(streams):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                p, xi, di = plans[i], x[offs[i]:offs[i] + splits[i]], dets[offs[i]:offs[i] + splits[i]]
                if stagger and i > 0 and mid_ev is not None:
                    st.wait_event(mid_ev)                      # lane i starts when lane i-1 is half way through the network
                eng.run_front(p, xi)
                mid = p.first_desc + (p.n_desc - p.first_desc) * 45 // 100
                seq(p, di, p.first_desc, mid)
                if stagger:
                    mid_ev = torch.cuda.Event(); mid_ev.record(st)
                seq(p, di, mid, p.n_desc)
            e = torch.cuda.Event(); e.record(st); main.wait_event(e)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

cases = [([B], False), ([B // 2, B - B // 2], False), ([B // 2, B - B // 2], True), ([3 * B // 4, B - 3 * B // 4], False),
         ([5 * B // 8, B - 5 * B // 8], False), ([B // 3, B // 3, B - 2 * (B // 3)], False), ([B // 4] * 4, False), ([B], False), ([B // 2, B - B // 2], False)]
for splits, stagger in cases:
    ms = run(splits, stagger)
    print("splits %-18s stagger=%d : %.3f ms per %d images -> %.0f img/s (convs only)" % (splits, stagger, ms, B, B / ms * 1e3)); sys.stdout.flush()
