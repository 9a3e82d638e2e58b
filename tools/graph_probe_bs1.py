"""bs=1 latency, eager vs HIP graph, per library build (YV3_LIB): python tools/graph_probe_bs1.py"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_v3_amd import synth, Detector
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
net = load_sw1_net(synth.weight_stream()).cuda()
x = torch.from_numpy(synth.images(1, 416, 5)).cuda()
for graph in (False, True, False, True):
    d = Detector(net, 1, 416, 416, graph=graph)
    for _ in range(10): r = d(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): r = d(x)
    dt = (time.perf_counter() - t0) / 100
    print("%s bs=1 graph=%d: %.3f ms per call" % (os.environ.get("YV3_LIB", "libyv3.so")[-16:], graph, dt * 1e3)); sys.stdout.flush()
