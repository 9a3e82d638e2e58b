#!/usr/bin/env python
"""Headline benchmark: YOLOv3 inference hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--size 416]

A step = one pass of the whole hot path over one batch of synthetic images that already live in
HBM: 75 convolutions (Darknet-53 + heads) -> 3-scale decode -> confidence filter -> per-class
greedy NMS -> final [B,cap,7] boxes copied to pinned host memory (asynchronously).  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank runs its own shard of the global
batch (weak scaling: --batch images PER GPU), and each step ends with the RCCL all-gather of the
final boxes.  Rank 0 prints ONE JSON line.

`roofline` is for the dominant kernel family, the fp32 implicit-GEMM convolution
(conv_igemm_f32_kernel, 74 launches per step): algorithmic FLOPs of those 74 convs for the batch
divided by the time of their launch sequence, measured with HIP events on the launch stream in every
timed step, against the 157.3 TFLOP/s fp32 MFMA peak.  `cpu_baseline` is the CPU oracle (torch fp32
CPU ops, same weights/inputs) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch                       # noqa: E402
import torch.distributed as dist   # noqa: E402

# MI355X_MICROARCH.md dense MFMA peaks.  f32x3: every fp32 product costs six bf16 MFMAs, so the ceiling
# for ALGORITHMIC fp32 FLOP/s in that mode is 2500/6 (frac == utilisation of the bf16 matrix pipe).
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f32x3": 2500.0 / 6, "f32h2": 2500.0 / 3}
DTYPE_NAME = {"f32": "f32 (exact fp32 MFMA)", "bf16": "bf16 (fp32 accumulate)",
              "f32x3": "f32 via exact 3-way bf16 split: 6 bf16 MFMAs per product, fp32 accumulate",
              "f32h2": "f32 via 2-way fp16 split (hi+lo): 3 fp16 MFMAs per product, fp32 accumulate"}
KERNEL_NAME = {"f32": "conv_igemm_f32_kernel", "bf16": "conv_planes_kernel<1>", "f32x3": "conv_planes_kernel<3>",
               "f32h2": "conv_planes_kernel<2>"}


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


def cpu_baseline(stream, size, n_img=8):
    """Oracle (CPU restatement of the reference path) on the host cores: forward + post-processing."""
    from oracle import oracle_cpu as oc
    from yolo_v3_amd import synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd, _ = oc.state_dict_from_stream(stream)
    x = torch.from_numpy(synth.images(n_img, size, 1))
    best = None
    t_all = time.perf_counter()
    for it in range(3):                                   # 1 warm-up + 2 timed, stop early if slow
        t0 = time.perf_counter()
        oc.detect(sd, x, 80, 0.5, 0.4)
        dt = time.perf_counter() - t0
        if it > 0:
            best = dt if best is None else min(best, dt)
        if time.perf_counter() - t_all > 25 and best is not None:
            break
    return {"value": round(n_img / best, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d images %dx%d, forward+decode+NMS, torch fp32 CPU ops, best of %d after 1 warm-up"
                      % (n_img, size, size, max(1, it))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--dtype", default="f32h2", choices=["f32", "f32x3", "f32h2", "bf16"],
                    help="conv math mode; f32h2, f32x3 and f32 all meet the 1e-4 fp32 parity bar (tests/test_gpu_e2e.py)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra exact-fp32-MFMA measurement")
    ap.add_argument("--conf", type=float, default=0.5)
    ap.add_argument("--nms", type=float, default=0.4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weights", default="sw1", choices=["sw1", "dense"],
                    help="sw1: ~50-150 candidates/img; dense: head biases raised so ~10^4 rows/img pass conf (BASELINE config 5)")
    args = ap.parse_args()

    from yolo_v3_amd import YoloNet, WeightManager, Detector, synth, arch, dist as ydist, _ffi

    rank, local, world = ydist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU path)"
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    codes = {"f32": _ffi.F32, "bf16": _ffi.BF16, "f32x3": _ffi.F32X3, "f32h2": _ffi.F32H2}

    # ---- model + data (synthetic SW-1 weights, synthetic scenes; both bit-reproducible)
    stream = synth.weight_stream() if args.weights == "sw1" else synth.dense_weight_stream()
    net = YoloNet((args.size, args.size)).eval()
    assert WeightManager(net).load_stream(stream) == stream.size
    net = net.to(dev)
    B = args.batch
    lo, _ = ydist.shard_range(B * world, rank, world)
    base = synth.images(min(B, 16), args.size, 1000 + lo)                 # 16 distinct scenes per rank, tiled
    x = torch.from_numpy(base).to(dev).repeat((B + base.shape[0] - 1) // base.shape[0], 1, 1, 1)[:B].contiguous()

    def measure(mode):
        det = Detector(net, B, args.size, args.size, args.conf, args.nms, dtype=codes[mode])
        eng, plan = det.engine, det.plan
        cap = det.pp.cap
        host_boxes = torch.empty((B * world, min(cap, 512), 7), dtype=torch.float32).pin_memory()
        host_counts = torch.empty((B * world,), dtype=torch.int32).pin_memory()

        conv_ev = []

        def step(timed):
            # conv section bracketed by events on the launch stream (torch's current stream)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lib = _ffi.lib()
            s = _ffi.stream_ptr()
            p0 = eng.packed[0]
            plan.bind_detections(det.dets)                       # fused decode: the head convs write the detections
            _ffi.check(lib.yv3_conv0(x.data_ptr(), p0.w.data_ptr(), p0.alpha.data_ptr(), p0.beta.data_ptr(),
                                     plan.conv0_out.data_ptr(), B, plan.H, plan.W, codes[mode], plan.flags.data_ptr(), s))
            if timed:
                e0.record()
            _ffi.check(lib.yv3_conv2d_sequence(plan.descs, plan.n_desc, s))
            if timed:
                e1.record()
                conv_ev.append((e0, e1))
            eng.run_decode(plan, det.dets)
            boxes, counts = det.pp.run_sync_free(det.dets, args.conf, args.nms, False, True, prob=True)
            kept = counts[B:]
            if world > 1:
                boxes, kept = ydist.gather_boxes(boxes[:, :host_boxes.shape[1]].contiguous(), kept)
            else:
                boxes = boxes[:, :host_boxes.shape[1]]
            host_boxes.copy_(boxes, non_blocking=True)
            host_counts.copy_(kept, non_blocking=True)

        def fence():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        with torch.no_grad():
            for _ in range(args.warmup):
                step(False)
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(True)
            fence()
            elapsed = time.perf_counter() - t0

        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        conv_ms = sum(a.elapsed_time(b) for a, b in conv_ev) / len(conv_ev)
        specs = arch.conv_specs()
        hw = arch.conv_output_hw(args.size)
        macs = [h * w * sp.cout * sp.cin * sp.k * sp.k for sp, (h, w) in zip(specs, hw)]
        flops_all = 2.0 * sum(macs) * B
        flops_igemm = 2.0 * sum(macs[1:]) * B                      # the 74 implicit-GEMM launches
        return elapsed, conv_ms, flops_all, flops_igemm, plan.n_desc, host_counts[:4].tolist()

    elapsed, conv_ms, flops_all, flops_igemm, n_desc, kept4 = measure(args.dtype)
    achieved = flops_igemm / (conv_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.dtype]
    secondary = None
    if args.dtype in ("f32x3", "f32h2") and not args.no_secondary:
        e2, c2, fa2, fi2, _, _ = measure("f32")
        a2 = fi2 / (c2 * 1e-3) / 1e12
        secondary = {"dtype": DTYPE_NAME["f32"], "value": round(B * world * args.steps / e2, 2), "unit": "images/sec",
                     "ms_per_step": round(e2 / args.steps * 1e3, 4),
                     "roofline": {"bound": "mfma", "kernel": KERNEL_NAME["f32"], "achieved": round(a2, 2),
                                  "peak": PEAK_TFLOPS["f32"], "unit": "TFLOP/s", "frac": round(a2 / PEAK_TFLOPS["f32"], 4)}}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        out = {
            "metric": "images/sec (YOLOv3 forward + decode + NMS, %dx%d, bs=%d per GPU)" % (args.size, args.size, B),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "ms_per_img": round(ms_per_step / B, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.dtype], "data": "synthetic",
            "config": {"workload": "%dx%d bs=%d per GPU, synthetic scenes, %s synthetic weights, conf=%.2f nms=%.2f"
                                   % (args.size, args.size, B, {"sw1": "SW-1", "dense": "SW-dense"}[args.weights], args.conf, args.nms),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "boxes_kept_first_images": kept4},
            "roofline": {"bound": "mfma", "kernel": "%s (74 launches/step)" % KERNEL_NAME[args.dtype],
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": None,
                         "conv_ms_per_step": round(conv_ms, 4), "launches": n_desc,
                         "avg_launch_ms": round(conv_ms / n_desc, 5),
                         "flop_per_launch_avg": flops_igemm / n_desc,
                         "end_to_end_frac": round(flops_all / (ms_per_step * 1e-3) / 1e12 / peak, 4)},
        }
        # HBM traffic of the dominant kernel family comes from separate rocprofv3 --pmc passes (FETCH_SIZE and
        # WRITE_SIZE cannot be sampled from inside this process); the committed summary for this exact
        # workload is attached when present (tools/traffic_summary.py, profiles/*_traffic_*.json).
        import glob
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic_%s_%d_bs%d.json" % (args.dtype, args.size, B))))
        tpath = cands[-1] if cands else ""
        if tpath:
            fam = json.load(open(tpath)).get(KERNEL_NAME[args.dtype].split("<")[0])
            if fam:
                out["roofline"]["traffic"] = round(fam["hbm_bytes_per_step_fetch_x2"] / n_desc)
                out["roofline"]["traffic_note"] = ("HBM bytes per launch (avg of %d launches/step) = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                                                   "from rocprofv3 --pmc passes of this workload; FETCH_SIZE doubled per the gfx950 "
                                                   "calibration in MI355X_MICROARCH.md (confirmed here on decode_kernel); source %s"
                                                   % (n_desc, os.path.basename(tpath)))
        if args.dtype in ("f32x3", "f32h2"):
            nm = {"f32x3": 6, "f32h2": 3}[args.dtype]
            out["roofline"]["note"] = ("achieved = algorithmic fp32 FLOP/s; peak = 2500 TFLOP/s dense 16-bit MFMA / %d MFMAs per "
                                       "fp32 product, so frac = matrix-pipe utilisation (%.0f TFLOP/s executed); the kernel is "
                                       "power-limited at ~1 PFLOP/s executed (DESIGN.md 3b)" % (nm, nm * achieved))
            # the same fp32-class result against the roofline of doing it with fp32 MFMAs (157.3 TFLOP/s dense)
            out["roofline"]["peak_fp32_mfma"] = PEAK_TFLOPS["f32"]
            out["roofline"]["frac_vs_fp32_mfma_peak"] = round(achieved / PEAK_TFLOPS["f32"], 4)
        if secondary is not None:
            out["exact_fp32_mfma"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(stream, args.size)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
