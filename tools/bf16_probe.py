import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from yolo_v3_amd import synth, _ffi, Detector
from oracle import oracle_cpu as oc
from tests.helpers import load_sw1_net, rel_err
torch.cuda.set_device(0)
stream = synth.weight_stream()
net = load_sw1_net(stream).cuda()
x = torch.from_numpy(synth.images(2, 416, 511)).cuda()
with torch.no_grad():
    d32 = net.forward_cat(x, dtype=_ffi.F32)
    dbf = net.forward_cat(x, dtype=_ffi.BF16)
    dh2 = net.forward_cat(x, dtype=_ffi.F32H2)
    dx3 = net.forward_cat(x, dtype=_ffi.F32X3)
sd, _ = oc.state_dict_from_stream(stream)
with torch.no_grad():
    ref = torch.cat(oc.yolonet_forward(sd, x.cpu()), 1)
for name, d in (("f32", d32), ("f32x3", dx3), ("f32h2", dh2), ("bf16", dbf)):
    e = rel_err(d.cpu(), ref)
    print("%-6s vs CPU oracle: max %.3g mean %.3g" % (name, float(e.max()), float(e.mean())))
e = rel_err(dbf.cpu(), d32.cpu())
from yolo_v3_amd import postprocessing
r32 = postprocessing(d32, 80, 0.5, 0.4); rbf = postprocessing(dbf, 80, 0.5, 0.4)
print("boxes f32", [len(r) for r in r32], "bf16", [len(r) for r in rbf])
