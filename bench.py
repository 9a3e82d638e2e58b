#!/usr/bin/env python
"""Headline benchmark: YOLOv3 inference hot path on MI355X (BASELINE.json metric: images/sec + ms/img at
416x416 bs=64 on 1/2/4/8 GPUs; NMS boxes delta vs ref).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64 | --global-batch G] [--size 416] [--dtype f32h2]

A step = one pass of the whole hot path over one batch of synthetic images already resident in HBM, run through
the PRODUCT entry point `Detector.run_device` (yolo_v3_amd/detect.py: conv0 -> 74 convs with the YOLO decode fused
into the head convs -> confidence filter -> per-class greedy NMS), then the final [B,cap,7] boxes + counts are
copied to pinned host memory asynchronously.  From ~12 images of 416x416 (8 of 608x608) the Detector runs the convolutions as two sub-batches
on two concurrent HIP streams ("lanes": same kernels and bits, they fill each other's idle CUs; --lanes 1 disables).  With N > 1 (torch.distributed.run, one rank per GPU) every rank runs
its shard and each step ends with the RCCL all-gather of the final boxes (yolo_v3_amd/dist.py); --batch is images PER
GPU (weak scaling, the default), --global-batch fixes the total (strong scaling; BASELINE configs[3] = 256 over 8).
Rank 0 prints ONE JSON line.

`roofline` is for the dominant kernel family, the implicit-GEMM convolution (conv_planes_kernel<2,...> in the default
fp16x2-plane mode: 71 launches per step behind the two fused front kernels -- first two layers, first residual block -- and 74 in the other modes): algorithmic FLOPs
(2*MAC) of those convs for the batch divided by the duration of their launch sequence, measured with HIP events on the
launch stream in every timed step (`all_75_convs_frac`: all 75 convs over front + convs time).  In the
default mode each fp32 product costs 3 fp16 MFMAs, so the peak for ALGORITHMIC FLOP/s is 2500/3 TFLOP/s and `frac`
is the utilisation of the 16-bit matrix pipe.  It is measured with ONE lane (the kernels alone on the chip: with two
concurrent lanes a kernel's duration includes the time it shares the chip, and rocprofv3's per-kernel durations sum to
~2x the wall time); `roofline.two_lanes_conv_section` is the rate of the conv section as the timed step runs it (all 75
convs' FLOPs over the fork -> join wall time).  `stages_ms` is the per-stage split of the timed step from HIP events.

Extra objects on the same line (rank 0, N = 1; --no-extras skips them):
  cpu_baseline   the CPU oracle (oracle/oracle_cpu.py: the reference path restated in torch fp32 CPU ops) timed on this
                 box's host cores on a bounded sample;
  boxes_delta    "NMS boxes delta vs ref": the HIP path's final boxes vs the oracle's on that same sample
                 (oracle/boxdelta.py: max rel error of coords / scores over matched boxes, class / count equality,
                 unmatched fraction);
  modes          the other fp32-class math modes on the headline workload (f32x3, exact f32);
  configs        BASELINE.json configs 1 / 2 / 4 (by list index) and eval mode at the reference's 0.005 / 0.45.
"""
import argparse
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch                       # noqa: E402
import torch.distributed as dist   # noqa: E402

# MI355X_MICROARCH.md dense MFMA peaks.  f32x3 / f32h2: every fp32 product costs six bf16 / three fp16 MFMAs, so the
# ceiling for ALGORITHMIC fp32 FLOP/s in those modes is 2500/6 and 2500/3 (frac == utilisation of the 16-bit matrix pipe).
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f32x3": 2500.0 / 6, "f32h2": 2500.0 / 3}
DTYPE_NAME = {"f32": "f32 (exact fp32 MFMA)", "bf16": "bf16 (fp32 accumulate)",
              "f32x3": "f32 via exact 3-way bf16 split: 6 bf16 MFMAs per product, fp32 accumulate",
              "f32h2": "f32 via 2-way fp16 split (hi+lo): 3 fp16 MFMAs per product, fp32 accumulate"}
KERNEL_NAME = {"f32": "conv_igemm_f32_kernel", "bf16": "conv_planes_kernel<1>", "f32x3": "conv_planes_kernel<3>",
               "f32h2": "conv_planes_kernel<2>"}
STAGES = ("conv0", "convs", "decode", "filter", "nms")


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(stream, size, conf, nms, n_img=8):
    """Oracle (CPU restatement of the reference path) on the host cores: forward + post-processing.
    Returns (cpu_baseline object, the sample images, the oracle's boxes for them)."""
    from oracle import oracle_cpu as oc
    from yolo_v3_amd import synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd, _ = oc.state_dict_from_stream(stream)
    x = torch.from_numpy(synth.images(n_img, size, 1))
    best, boxes = None, None
    t_all = time.perf_counter()
    for it in range(3):                                   # 1 warm-up + 2 timed, stop early if slow
        t0 = time.perf_counter()
        boxes = oc.detect(sd, x, 80, conf, nms)
        dt = time.perf_counter() - t0
        if it > 0:
            best = dt if best is None else min(best, dt)
        if time.perf_counter() - t_all > 25 and best is not None:
            break
    obj = {"value": round(n_img / best, 3), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": "%d images %dx%d, forward+decode+NMS, torch fp32 CPU ops, best of %d after 1 warm-up"
                     % (n_img, size, size, max(1, it))}
    return obj, x, boxes


class Workload:
    """One (weights, batch, size, mode, thresholds) configuration measured through Detector.run_device."""

    def __init__(self, net, x, mode, conf, nms, is_eval=False, world=1, cap_host=512, max_cand=None, lanes=None):
        from yolo_v3_amd import Detector, _ffi
        codes = {"f32": _ffi.F32, "bf16": _ffi.BF16, "f32x3": _ffi.F32X3, "f32h2": _ffi.F32H2}
        self.x, self.world, self.mode = x, world, mode
        B, _, H, W = x.shape
        self.B, self.size = B, H
        self.det = Detector(net, B, H, W, conf, nms, is_eval=is_eval, dtype=codes[mode], max_cand=max_cand, lanes=lanes)
        self.cap_host = min(self.det.pp.cap, cap_host)
        self.host_boxes = torch.empty((B * world, self.cap_host, 7), dtype=torch.float32).pin_memory()
        self.host_counts = torch.empty((2 * B * world,), dtype=torch.int32).pin_memory()
        self.events = []

    def step(self, timed):
        from yolo_v3_amd import dist as ydist
        marks = {}

        def mark(name):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                                   # on torch's current stream == the kernels' launch stream
            marks[name] = ev

        boxes, counts = self.det.run_device(self.x, mark if timed else None)     # product code: conv0 ... NMS
        boxes = boxes[:, :self.cap_host]
        if self.world > 1:
            boxes, counts = ydist.gather_boxes(boxes.contiguous(), counts)
            if timed:
                mark("gather")
        self.host_boxes.copy_(boxes, non_blocking=True)
        self.host_counts.copy_(counts, non_blocking=True)
        if timed:
            mark("d2h")
            self.events.append(marks)

    def run(self, steps, warmup):
        def fence():
            if self.world > 1:
                dist.barrier()
            torch.cuda.synchronize()
        with torch.no_grad():
            for _ in range(warmup):
                self.step(False)
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(True)
            fence()
            elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.x.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # status word of the default mode (fp16 saturation) -- product code checks it in Detector.__call__
        self.det.engine.raise_if_overflowed(self.det.plan, int(self.det.plan.flags.item()))
        return elapsed

    def stages_ms(self):
        order = ["start"] + [s for s in STAGES] + (["gather"] if self.world > 1 else []) + ["d2h"]
        out = {}
        for prev, cur in zip(order, order[1:]):
            out[cur] = round(sum(m[prev].elapsed_time(m[cur]) for m in self.events) / len(self.events), 4)
        return out

    def flops(self):
        """(2*MAC of all 75 convs, 2*MAC of the launches timed as the 'convs' stage, number of those launches).
        The 'conv0' stage is the network's front: feature.mlist.0 alone, or -- fp16-plane mode, csrc/conv_front.hip -- the
        first TWO layers in one launch; the 'convs' stage is the remaining 74 / 73 implicit-GEMM launches."""
        from yolo_v3_amd import arch
        specs = arch.conv_specs()
        hw = arch.conv_output_hw(self.size)
        macs = [h * w * sp.cout * sp.cin * sp.k * sp.k for sp, (h, w) in zip(specs, hw)]
        plan = self.det.plan
        first = 1 + plan.first_desc
        if self.det.lanes > 1:                              # lanes: the 'convs' stage covers every conv launch of every lane
            return 2.0 * sum(macs) * self.B, 2.0 * sum(macs) * self.B, sum(1 + (p.first_desc > 0) + (p.first_desc > 1) + p.n_desc - p.first_desc
                                                                         for p in self.det.lane_plans)
        return 2.0 * sum(macs) * self.B, 2.0 * sum(macs[first:]) * self.B, plan.n_desc - plan.first_desc

    def summary(self, elapsed, steps):
        st = self.stages_ms()
        fa, fi, nl = self.flops()
        ach = fi / (st["convs"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[self.mode]
        return {"dtype": DTYPE_NAME[self.mode], "value": round(self.B * self.world * steps / elapsed, 2), "unit": "images/sec",
                "ms_per_step": round(elapsed / steps * 1e3, 4), "ms_per_img": round(elapsed / steps * 1e3 / (self.B * self.world), 5),
                "stages_ms": st,
                "lanes": self.det.lanes,
                "roofline": {"bound": "mfma", "kernel": ("%s (%d launches/step)" % (KERNEL_NAME[self.mode], nl)) if self.det.lanes == 1 else
                             ("all conv launches of %d concurrent lanes (%d/step): FLOPs over the wall time of the conv section" % (self.det.lanes, nl)),
                             "achieved": round(ach, 2),
                             "peak": round(peak, 2), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "launches": nl,
                             "all_75_convs_frac": round(fa / ((st["conv0"] + st["convs"]) * 1e-3) / 1e12 / peak, 4)}}


def make_net(stream, size, dev):
    from yolo_v3_amd import YoloNet, WeightManager
    net = YoloNet((size, size)).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    return net.to(dev)


def scenes(B, size, seed, dev, distinct=16):
    from yolo_v3_amd import synth
    base = synth.images(min(B, distinct), size, seed)                     # `distinct` different scenes, tiled
    return torch.from_numpy(base).to(dev).repeat((B + base.shape[0] - 1) // base.shape[0], 1, 1, 1)[:B].contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="total images per step, split over the GPUs (strong scaling)")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--dtype", default="f32h2", choices=["f32", "f32x3", "f32h2", "bf16"],
                    help="conv math mode; f32h2, f32x3 and f32 all meet the 1e-4 fp32 parity bar (tests/test_gpu_e2e.py)")
    ap.add_argument("--conf", type=float, default=0.5)
    ap.add_argument("--nms", type=float, default=0.4)
    ap.add_argument("--lanes", type=int, default=0, help="sub-batches run concurrently on separate HIP streams (0 = Detector's default: "
                    "2 when it measures a gain on this GPU, else 1)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra modes / configs measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weights", default="sw1", choices=["sw1", "dense", "eval"],
                    help="sw1: ~50-150 candidates/img; dense: ~1e4 rows/img pass conf (BASELINE configs[4]); eval: SW-eval")
    args = ap.parse_args()

    from yolo_v3_amd import synth, dist as ydist

    rank, local, world = ydist.init_from_env()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU path)"
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    strong = args.global_batch > 0
    if strong:
        assert args.global_batch % world == 0, "--global-batch must be a multiple of the number of GPUs"
        B = args.global_batch // world
    else:
        B = args.batch
    streams = {"sw1": synth.weight_stream, "dense": synth.dense_weight_stream, "eval": synth.eval_weight_stream}
    stream = streams[args.weights]()
    net = make_net(stream, args.size, dev)
    lo, _ = ydist.shard_range(B * world, rank, world)
    x = scenes(B, args.size, 1000 + lo, dev)

    main_w = Workload(net, x, args.dtype, args.conf, args.nms, world=world, lanes=args.lanes or None)
    elapsed = main_w.run(args.steps, args.warmup)
    if rank == 0 and os.environ.get("YV3_DUMP_PLAN"):               # for tools/trace_layers.py: conv spec index of every launch
        p_ = main_w.det.plan
        json.dump({"first_desc": p_.first_desc, "desc_spec": p_.desc_spec}, open(os.environ["YV3_DUMP_PLAN"], "w"))
    head = main_w.summary(elapsed, args.steps)
    kept4 = main_w.host_counts[B:B + 4].tolist()                       # rank 0's shard: [0:B] candidates, [B:2B] kept

    # The roofline of the dominant KERNEL is measured with the kernels running alone (one lane): with two concurrent lanes a
    # kernel's duration includes the time it shares the chip with the other lane's kernels (rocprofv3 then shows per-kernel
    # durations that sum to ~2x the wall time).  The two-lane conv-section rate (FLOPs over wall) is reported next to it.
    lanes_used = main_w.det.lanes
    roof_w, head1 = main_w, head
    if lanes_used > 1 and rank == 0:
        # rank-local (world=1: no collective inside) -- whether a rank runs two lanes is ITS calibration's decision, so a pass
        # with barriers here could leave the ranks waiting for each other
        roof_w = Workload(net, x, args.dtype, args.conf, args.nms, world=1, lanes=1)
        e1 = roof_w.run(min(args.steps, 10), 3)
        head1 = roof_w.summary(e1, min(args.steps, 10))
    out = None
    if rank == 0:
        fa, fi, n_desc = roof_w.flops()
        st = head1["stages_ms"]
        out = {
            "metric": "images/sec (YOLOv3 forward + decode + NMS, %dx%d, bs=%d per GPU)" % (args.size, args.size, B),
            "value": head["value"], "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "ms_per_img": head["ms_per_img"],
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.dtype], "data": "synthetic",
            "config": {"workload": "%dx%d bs=%d per GPU, synthetic scenes, %s synthetic weights, conf=%.2f nms=%.2f"
                                   % (args.size, args.size, B, {"sw1": "SW-1", "dense": "SW-dense", "eval": "SW-eval"}[args.weights],
                                      args.conf, args.nms),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "entry": "Detector.run_device",
                       "boxes_kept_first_images": kept4},
            "stages_ms": head["stages_ms"], "lanes": lanes_used,
            "roofline": dict(head1["roofline"], traffic=None,
                             conv_ms_per_step=st["convs"], avg_launch_ms=round(st["convs"] / n_desc, 5),
                             flop_per_launch_avg=fi / n_desc,
                             end_to_end_frac=round(fa / (head["ms_per_step"] * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4)),
        }
        if lanes_used > 1:
            out["roofline"]["measured_with"] = ("lanes=1 (%.2f images/s, %.3f ms/step): the kernels run alone, so HIP-event and rocprofv3 per-kernel durations "
                                                "mean what they say; the timed step above runs %d concurrent lanes" % (head1["value"], head1["ms_per_step"], lanes_used))
            out["roofline"]["two_lanes_conv_section"] = {k: head["roofline"][k] for k in ("kernel", "achieved", "frac", "launches")}
            out["lanes_calibration_ms"] = getattr(main_w.det, "lane_calibration", None)
        # HBM traffic of the dominant kernel family: FETCH_SIZE / WRITE_SIZE need their own rocprofv3 --pmc passes
        # (they cannot be sampled from inside this process); the committed summary of those passes over THIS workload
        # is attached and labelled with its source run (tools/traffic_summary.py -> profiles/*_traffic_*.json).
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic_%s_%d_bs%d.json" % (args.dtype, args.size, B))))
        if cands:
            fam = json.load(open(cands[-1])).get(KERNEL_NAME[args.dtype].split("<")[0])
            if fam:
                out["roofline"]["traffic"] = round(fam["hbm_bytes_per_step_fetch_x2"] / n_desc)
                out["roofline"]["traffic_source"] = os.path.basename(cands[-1])
                out["roofline"]["traffic_note"] = ("HBM bytes per launch (avg of %d launches/step) = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate "
                                                   "rocprofv3 --pmc passes of this workload, NOT measured in this run; FETCH_SIZE doubled per the "
                                                   "gfx950 calibration in MI355X_MICROARCH.md" % n_desc)
        if args.dtype in ("f32x3", "f32h2"):
            nm = {"f32x3": 6, "f32h2": 3}[args.dtype]
            ach = out["roofline"]["achieved"]
            out["roofline"]["note"] = ("achieved = algorithmic fp32 FLOP/s; peak = 2500 TFLOP/s dense 16-bit MFMA / %d MFMAs per fp32 "
                                       "product, so frac = matrix-pipe utilisation (%.0f TFLOP/s executed)" % (nm, nm * ach))
            out["roofline"]["frac_vs_fp32_mfma_peak"] = round(ach / PEAK_TFLOPS["f32"], 4)

    extras = world == 1 and not args.no_extras and rank == 0
    if extras:
        sub_steps, sub_warm = 10, 3
        # ---- the other fp32-class modes on the headline workload (driver-timed, same entry point)
        out["modes"] = {}
        # (+ bf16: REDUCED precision -- conv operands rounded to bfloat16 as in BASELINE configs[2]; not a parity mode)
        for mode in ("f32x3", "f32", "bf16"):
            if mode == args.dtype:
                continue
            w = Workload(net, x, mode, args.conf, args.nms)
            out["modes"][mode] = w.summary(w.run(sub_steps, sub_warm), sub_steps)
            if mode == "bf16":
                out["modes"][mode]["note"] = "reduced precision (bf16 conv operands, fp32 accumulate / epilogue / decode): outside the 1e-4 parity bar"
            del w
        # ---- BASELINE.json configs (list indices): 1 = 416 bs32 fp32; 2 = 608 bs16 bf16 convs; 4 = 608 bs8 dense scene
        out["configs"] = {}

        def sub(key, label, net_, x_, mode, conf, nms, **kw):
            w = Workload(net_, x_, mode, conf, nms, **kw)
            s = w.summary(w.run(sub_steps, sub_warm), sub_steps)
            B_ = x_.shape[0]
            s["workload"] = label
            s["kept_per_img_first4"] = w.host_counts[B_:B_ + 4].tolist()
            s["candidates_per_img_first4"] = w.host_counts[:4].tolist()
            out["configs"][key] = s

        sub("1", "416x416 bs=32 SW-1 fp32-class (f32h2) conf=0.5 nms=0.4", net, scenes(32, 416, 1, dev, 32), "f32h2", 0.5, 0.4)
        net608 = make_net(stream, 608, dev)
        sub("2", "608x608 bs=16 SW-1 bf16 convs / fp32 decode conf=0.5 nms=0.4", net608, scenes(16, 608, 2, dev), "bf16", 0.5, 0.4)
        del net608
        dnet = make_net(synth.dense_weight_stream(), 608, dev)
        sub("4", "608x608 bs=8 SW-dense (>=5k pre-NMS rows/img) f32h2 conf=0.5 nms=0.4", dnet, scenes(8, 608, 4, dev, 8), "f32h2", 0.5, 0.4,
            cap_host=8192)
        del dnet
        enet = make_net(synth.eval_weight_stream(), 416, dev)
        sub("eval", "416x416 bs=32 SW-eval, eval mode as evaluate.py:201-204 (conf=0.005 nms=0.45 is_eval=True)", enet,
            scenes(32, 416, 5, dev, 32), "f32h2", 0.005, 0.45, is_eval=True, cap_host=4096, max_cand=8192)
        del enet
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.boxdelta import boxes_delta
        from yolo_v3_amd import detect
        cb, xs, ref_boxes = cpu_baseline(stream, args.size, args.conf, args.nms)
        out["cpu_baseline"] = cb
        # NMS boxes delta vs ref: the product entry (detect) on the SAME sample the oracle just ran
        from yolo_v3_amd import _ffi
        net.math_mode = {"f32": _ffi.F32, "bf16": _ffi.BF16, "f32x3": _ffi.F32X3, "f32h2": _ffi.F32H2}[args.dtype]
        got = detect(net, xs.to(dev), 80, args.conf, args.nms)
        d = boxes_delta(got, ref_boxes, xs.shape[0])
        out["boxes_delta"] = {
            "vs": "CPU oracle (reference path restated, oracle/oracle_cpu.py) on the cpu_baseline sample", "images": d["images"],
            "ref_boxes": d["ref_boxes"], "got_boxes": d["got_boxes"], "matched_iou_ge_0.999": d["matched"],
            "unmatched_frac": round(d["unmatched_frac"], 6), "count_equal_images": d["count_equal_images"],
            "class_equal_images": d["class_equal_images"], "max_rel_err_coords": float("%.3g" % d["max_rel_err_coords"]),
            "max_abs_err_conf": float("%.3g" % d["max_abs_err_conf"]), "max_abs_err_score": float("%.3g" % d["max_abs_err_score"]),
            "tolerance": 1e-4}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
