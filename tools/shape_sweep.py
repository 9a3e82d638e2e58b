"""Ad-hoc robustness sweep: whole-net detections vs the CPU oracle over odd batch sizes / image sizes / schedules."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v3_amd import synth, _ffi
from oracle import oracle_cpu as oc
from tests.helpers import load_sw1_net
torch.cuda.set_device(0)
stream = synth.weight_stream()
sd, _ = oc.state_dict_from_stream(stream)
worst = 0.0
MODE = {"f32h2": _ffi.F32H2, "f32": _ffi.F32, "f32x3": _ffi.F32X3}[os.environ.get("DT", "f32h2")]      # DT: math mode to sweep
for sk in ((False, True) if MODE == _ffi.F32H2 else (False,)):
    net = load_sw1_net(stream).cuda()
    net.math_mode = MODE
    net.stream_k = sk
    for (B, H, W) in [(1, 416, 416), (3, 320, 320), (5, 352, 608), (7, 608, 352), (2, 96, 64), (9, 224, 416), (17, 256, 256), (33, 160, 192)]:
        net.img_dim = (W, H)
        x = torch.from_numpy(synth.images(B, max(H, W), 100 + B)[:, :, :H, :W].copy())
        with torch.no_grad():
            got = net.forward_cat(x.cuda()).cpu()
            ref = torch.cat(oc.yolonet_forward(sd, x), 1)
        err = ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
        worst = max(worst, err)
        print("stream_k=%d B=%d %dx%d: max normalised error %.3g %s" % (sk, B, H, W, err, "" if err < 1e-4 else "  <-- FAIL")); sys.stdout.flush()
print("worst", worst)
