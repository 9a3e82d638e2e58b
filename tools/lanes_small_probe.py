import sys, os
sys.path.insert(0, os.getcwd())
import torch
from yolo_v3_amd import synth, Detector
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
for size, Bs in ((416, (2, 4, 8, 12, 16)), (608, (2, 4, 8))):
    net = load_sw1_net(synth.weight_stream(), size).cuda()
    for B in Bs:
        d = Detector(net, B, size, size, lanes=2)
        d._calibrate_lanes()
        print(size, B, d.lane_calibration, "-> lanes", d.lanes)
