"""bench.py's own N-rank launch path, on CPU: `python bench.py --gpus N` (no torchrun environment) must become N ranks,
run the product's pack -> ONE all-gather -> assemble composition (--dry-run: CPU tensors over gloo, no kernels) and print one
JSON line with n_gpus == N; without enough GPUs a real run must fail loudly, not fall back to one rank."""
import json
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, cwd=REPO, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_self_launch_two_ranks_dry_run():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ok"] is True and out["config"]["parallelism"] == "dp2"
    assert out["config"]["global_batch"] == 8 and out["config"]["backend"] == "gloo"


def test_one_rank_dry_run():
    r = _run(["--dry-run", "--steps", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 4096                                  # the driver's parser gave up on a 22 KB line (BENCH_r04.parsed == null)
    assert json.loads(last)["n_gpus"] == 1


def test_final_line_is_compact_for_a_full_result_object():
    """bench.compact_line over a FULL result object of a real run (profiles/r04zy_bench.json, 22 KB: modes, configs, PCIe passes,
    notes): the line the driver parses stays < 4 KB, is valid JSON, keeps the contract's keys and carries no note strings."""
    import bench
    full = json.load(open(os.path.join(REPO, "profiles", "r04zy_bench.json")))
    full["roofline"]["algorithmic_bytes"] = 264000000
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json"), separators=(",", ":"))
    assert len(line) < 4096, len(line)
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "boxes_delta", "exact_f32"):
        assert k in c, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "executed_frac", "mfma_util", "algorithmic_bytes"):
        assert k in c["roofline"], k
    assert set(c["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert "model" not in c["config"] and "workload" in c["config"]
    assert not any("note" in k or k.endswith("_is") for k in c["roofline"])


def test_final_line_of_the_round6_object_names_a_reference_precision_mode():
    """VERDICT r5 #1 on a FULL result object of a round-6 run (profiles/r06z_bench_full.json): the compact line's top-level dtype is the
    exact-fp32 mode, `roofline.frac` is the EXECUTED fraction (<= 1) with the algorithmic one beside it, the kernel family / traffic /
    byte counts are there, and the product's default mode, f32x3 and bf16 are sub-objects with the same fields; still < 4 KB."""
    import bench
    full = json.load(open(os.path.join(REPO, "profiles", "r06z_bench_full.json")))
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json"), separators=(",", ":"))
    assert len(line) < 4096, len(line)
    c = json.loads(line)
    assert c["dtype"].startswith("f32 (exact fp32") and c["config"]["workload"].startswith("416x416 bs=64")
    r = c["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 157.3 and 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_frac"] > r["frac"] and r["winograd4_launches"] == 31 and "conv_wino4_f32_kernel" in r["kernel"]
    assert r["traffic"] > r["form_bytes"] > r["algorithmic_bytes"] > 0 and 0.0 < r["mfma_util"] <= 1.0
    for k in ("f32h2", "f32x3", "bf16"):
        assert {"value", "ms_per_step", "achieved", "peak", "frac", "algorithmic_frac", "launches"} <= set(c[k]), (k, c[k])
        assert c[k]["frac"] <= 1.0
    assert c["configs"]["1"]["value"] > 0 and "1_f32h2" in c["configs"]
    assert c["cpu_baseline"]["kind"] == "port" and c["boxes_delta"]["matched"] == c["boxes_delta"]["ref"]


def test_too_few_gpus_is_an_error_not_a_one_rank_run():
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(have + 2), "--no-extras", "--steps", "1"])
    assert r.returncode == 2 and "visible GPUs" in r.stderr and "{" not in r.stdout


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_live_traffic_counter_parsing(tmp_path):
    """bench.live_traffic sums a rocprofv3 counter_collection.csv per kernel family: per dispatch (a counter can appear once per
    XCD / shader engine for the same dispatch), only the family's kernels."""
    import bench
    p = tmp_path / "t_counter_collection.csv"
    p.write_text('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
                 '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value"\n'
                 '1,1,0,1,9,9,512,3,"void (anonymous namespace)::conv_planes_kernel<2, 256, 128>(ConvParamsP)",512,0,0,128,0,96,"FETCH_SIZE",100.5\n'
                 '1,1,0,1,9,9,512,3,"void (anonymous namespace)::conv_planes_kernel<2, 256, 128>(ConvParamsP)",512,0,0,128,0,96,"FETCH_SIZE",20\n'
                 '2,2,0,1,9,9,512,4,"(anonymous namespace)::mask_kernel(int const*)",256,0,0,34,0,60,"FETCH_SIZE",7\n'
                 '3,3,0,1,9,9,512,3,"void (anonymous namespace)::conv_planes_kernel<2, 128, 128>(ConvParamsP)",512,0,0,128,0,96,"FETCH_SIZE",30\n')
    total, n = bench.sum_counter(str(p), "conv_planes_kernel")
    assert n == 2 and abs(total - 150.5) < 1e-9
    assert bench.sum_counter(str(p), "conv_igemm_f32_kernel") == (0, 0)
