"""The hostile calibrated network (tests/helpers.hostile_state_dict) in the exact-fp32 mode with every eligible 3x3 layer forced into one
form -- direct / Winograd F(2x2,3x3) / F(4x4,3x3) -- against an fp64 evaluation of the oracle (head logits) and the fp32 oracle itself
(detections): the GPU side of tools/winograd_f32_gate.py.   python tools/hostile_forms.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle_cpu as oc
from yolo_v3_amd import YoloNet, WeightManager, _ffi
from tests.helpers import hostile_state_dict, state_dict_to_stream, rel_err
torch.cuda.set_device(0)
sd, x = hostile_state_dict()
stream = state_dict_to_stream(sd)
with torch.no_grad():
    l64 = oc.head_logits({k: v.double() for k, v in sd.items()}, x.double())
    l32 = oc.head_logits(sd, x)
    d32 = torch.cat(oc.yolonet_forward(sd, x), 1)
print("fp32 CPU oracle: logits vs fp64 %.3g" % max(float(rel_err(a, b).max()) for a, b in zip(l32, l64)))
for name, wino, w4 in (("direct", False, False), ("F(2x2,3x3) on 31 layers", "always", False), ("F(4x4,3x3) on 31 layers", "always", True)):
    net = YoloNet((416, 416)).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    net = net.cuda()
    net.math_mode, net.winograd, net.winograd4 = _ffi.F32, wino, w4
    eng = net.engine()
    eng.fuse_decode, eng._plans = False, {}
    with torch.no_grad():
        _, plan = eng.forward(x.cuda())
        torch.cuda.synchronize()
        lg = [t.permute(0, 3, 1, 2).float().cpu() for (t, _, _) in plan.logits]
        forms = [f for _, f in plan.forms()]
        eng.fuse_decode, eng._plans = True, {}
        dets = net.forward_cat(x.cuda()).cpu()
    ok = torch.isfinite(d32) & (d32.abs() < 1e30)
    print("%-26s forms %s: logits vs fp64 %.3g, detections vs fp32 oracle %.3g" % (name, {f: forms.count(f) for f in set(forms)},
          max(float(rel_err(a, b).max()) for a, b in zip(lg, l64)), float(rel_err(dets[ok], d32[ok]).max())))
