"""MFMA-pipe utilisation / clock per kernel family from ONE rocprofv3 pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d DIR -o t -- python bench.py ...
(counters in their own run, kernel-trace only).  Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed
over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE counts cycles per XCD, summed over the 8 XCDs; effective clock =
GRBM_GUI_ACTIVE / 8 / kernel wall time.
    python tools/mfma_util_summary.py DIR/<host>/<pid> t [B size] > profiles/rNN_mfma_util.json"""
import collections
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    assert hits, "no %s under %s" % (suffix, d)
    return hits[0]


def main():
    d = sys.argv[1]
    rows = list(csv.DictReader(open(find(d, "_counter_collection.csv"))))
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    dur = {int(r["Dispatch_Id"]): int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(find(d, "_kernel_trace.csv")))}
    fam = collections.OrderedDict()
    for di, c in disp.items():
        n = c["name"]
        targs = [a.strip() for a in n.split("conv_planes_kernel<")[1].split(">")[0].split(",")] if "conv_planes_kernel<" in n else []
        # template <NP, BM, BN, WM, WN, NSTAGE, K3, DUAL, OUT_F32, PP, SK, MINW, MTG, WINO, ROLL>
        wino = len(targs) > 13 and targs[13] == "true"
        roll = len(targs) > 14 and targs[14] == "true"
        tile = ",".join(targs[:5]) if targs else ""
        key = ("wino_input_kernel (Winograd input transform)" if "wino_input" in n else
               "conv_planes_w4_kernel (fp16 planes, four-wave 192x128 tile, two workgroups per CU)" if "conv_planes_w4_kernel" in n else
               "conv_planes_kernel<2,128,128,...,WINO> (Winograd GEMM stage)" if wino else
               "conv_planes_kernel<1,256,256,2,4,ROLL> (bf16, 8 waves, 128x64 wave tiles, rolling loop)" if tile == "1,256,256,2,4" else
               "conv_planes_kernel<1,256,128,2,2,ROLL> (bf16, four waves, rolling loop)" if tile == "1,256,128,2,2" and roll else
               "conv_planes_kernel<1,256,128,2,2> (bf16, four waves, 128x64 wave tiles)" if tile == "1,256,128,2,2" else
               "conv_planes_kernel<1,256,128,4,2> (bf16, 8-wave ping-pong)" if tile == "1,256,128,4,2" else
               "conv_planes_kernel<2,256,128> (3x3 / 1x1, 8-wave ping-pong)" if tile.startswith("2,256,128") else
               "conv_planes_kernel<2,*> other tiles" if targs and targs[0] == "2" else
               "conv_planes_kernel (other modes)" if "conv_planes" in n else
               "conv_front" if "front" in n else "conv_res64" if "conv_res64" in n else "conv_1x1" if "conv1x1" in n else
               "conv_wino4_f32_kernel (exact fp32, Winograd F(4x4,3x3) GEMM stage + fold)" if "conv_wino4" in n else
               "wino4_input_f32_kernel (F(4x4,3x3) input transform)" if "wino4_input" in n else
               "conv_gemm1x1_f32_kernel (exact fp32, plain 1x1 layers, persistent DMA-fed GEMM)" if "conv_gemm1x1" in n else
               "conv_igemm_f32_kernel" if "conv_igemm" in n else "conv0" if "conv0" in n else
               "postproc" if any(k in n for k in ("filter_kernel", "rank_kernel", "rank_seg_kernel", "segpart_kernel", "mask_kernel", "scan_kernel", "compact_kernel", "zero_kernel")) else None)
        if key is None or di not in dur:
            continue
        f = fam.setdefault(key, {"dispatches": 0, "wall_ns": 0, "mfma_busy": 0.0, "sq_busy": 0.0, "gui": 0.0})
        f["dispatches"] += 1
        f["wall_ns"] += dur[di]
        f["mfma_busy"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        f["sq_busy"] += c.get("SQ_BUSY_CYCLES", 0.0)
        f["gui"] += c.get("GRBM_GUI_ACTIVE", 0.0)
    out = {"source": os.path.abspath(d), "note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); "
           "clock_ghz = GRBM_GUI_ACTIVE/8 / wall; profiled passes clock ~3 % lower than un-profiled ones (guide: DVFS give-back)",
           "families": {}}
    for key, f in fam.items():
        cyc = f["gui"] / 8.0
        out["families"][key] = {"dispatches": f["dispatches"], "wall_ms_total": round(f["wall_ns"] * 1e-6, 4),
                                "clock_ghz": round(cyc / f["wall_ns"], 3) if f["wall_ns"] else None,
                                "mfma_util": round(f["mfma_busy"] / (cyc * 1024), 4) if cyc else None,
                                "mfma_busy_tflops_fp16_equiv": round(f["mfma_busy"] / 32 * 32 * 32 * 16 * 2 / (f["wall_ns"] * 1e-9) / 1e12, 1) if f["wall_ns"] else None,
                                "sq_busy_frac": round(f["sq_busy"] / (f["gui"] * 4) , 4) if f["gui"] else None}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
