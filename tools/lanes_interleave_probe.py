"""Does the ORDER in which the host enqueues the two lanes matter?  Default: lane 0's whole launch sequence, then lane 1's
(lane 0 gets a ~0.3 ms head start).  Here: the sequences enqueued in alternating slices of SL descriptors."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import synth, Detector, _ffi
from yolo_v3_amd._ffi import ConvDesc
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
net = load_sw1_net(synth.weight_stream()).cuda()
B = 64
x = torch.from_numpy(synth.images(16, 416, 7)).cuda().repeat(4, 1, 1, 1).contiguous()
det = Detector(net, B, 416, 416, lanes=None)
print("lanes", det.lanes, getattr(det, "lane_calibration", None))
lib = _ffi.lib()

def run_interleaved(SL):
    main = torch.cuda.current_stream()
    fork = torch.cuda.Event(); fork.record(main)
    eng = det.engine
    for p, off, st in zip(det.lane_plans, det.lane_off, det.lane_streams):
        st.wait_event(fork)
        with torch.cuda.stream(st):
            eng.run_front(p, x[off:off + p.B])
            p.bind_detections(det.dets[off:off + p.B])
    n = det.lane_plans[0].n_desc; first = det.lane_plans[0].first_desc
    for s0 in range(first, n, SL):
        for p, st in zip(det.lane_plans, det.lane_streams):
            with torch.cuda.stream(st):
                tail = ctypes.cast(ctypes.addressof(p.descs) + s0 * ctypes.sizeof(ConvDesc), ctypes.POINTER(ConvDesc))
                _ffi.check(lib.yv3_conv2d_sequence(tail, min(SL, n - s0), _ffi.stream_ptr()), "seq")
    for st in det.lane_streams:
        done = torch.cuda.Event(); done.record(st); main.wait_event(done)

def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

noop = lambda name: None
for rep in range(2):
    print("default order      : %.3f ms" % timed(lambda: det._run_convs(x, noop)))
    for SL in (1, 4, 12, 36):
        print("interleaved SL=%-3d : %.3f ms" % (SL, timed(lambda: run_interleaved(SL))))
ref = det.dets.clone(); det._run_convs(x, noop); torch.cuda.synchronize()
print("same bits:", torch.equal(ref, det.dets))
