import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_v3_amd import synth, Detector
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
for name, stream, args in (("eval bs32", synth.eval_weight_stream(), dict(obj_conf_thr=0.005, nms_thr=0.45, is_eval=True)),
                           ("sw1 bs32", synth.weight_stream(), {}), ("sw1 bs8", synth.weight_stream(), {})):
    B = 8 if "bs8" in name else 32
    net = load_sw1_net(stream).cuda()
    x = torch.from_numpy(synth.images(B, 416, 5)).cuda()
    for graph in (False, True, False, True):
        d = Detector(net, B, 416, 416, graph=graph, **args)
        for _ in range(5): r = d(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): r = d(x)
        dt = (time.perf_counter() - t0) / 30
        print("%-10s graph=%d lanes=%d: %.3f ms per call (sync + list each call), %.0f img/s" % (name, graph, d.lanes, dt * 1e3, B / dt)); sys.stdout.flush()
