"""How many host threads should the CPU baseline use on this box?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle_cpu as oc
from yolo_v3_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null")
sd, _ = oc.state_dict_from_stream(synth.weight_stream())
x = torch.from_numpy(synth.images(8, 416, 1))
for t in (8, 16, 32, 64, 128):
    torch.set_num_threads(t)
    with torch.no_grad():
        oc.yolonet_forward(sd, x[:2])
        t0 = time.perf_counter(); oc.yolonet_forward(sd, x); dt = time.perf_counter() - t0
    print("threads", t, "8 img fwd %.2fs -> %.2f img/s" % (dt, 8 / dt)); sys.stdout.flush()
