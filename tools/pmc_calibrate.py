"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB) for the conv kernels' staging instruction: tools/probes/hbm_stream streams a
2 GiB buffer (8x the Infinity Cache) with global_load_lds_dwordx4 -- every full launch reads exactly 2 097 152 KiB (its warm-up launches an
eighth) and, in the second table, writes half of that with 16-byte stores.  python tools/pmc_calibrate.py DIR_FETCH DIR_WRITE"""
import csv, glob, os, sys, collections
def per_dispatch(d):
    hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = collections.OrderedDict()
    for r in csv.DictReader(open(hits[0])):
        if "stream_kernel" in r["Kernel_Name"]:
            e = per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0, int(r["Grid_Size"])])
            e[1] += float(r["Counter_Value"])
    return per
for label, d, full in (("FETCH_SIZE", sys.argv[1], 2097152.0), ("WRITE_SIZE", sys.argv[2], 1048576.0)):
    per = per_dispatch(d)
    ratios = {}
    for name, v, grid in per.values():
        wr = "true" in name.split("<")[1]
        if label == "WRITE_SIZE" and not wr:
            continue
        # a full launch follows its 1/8 warm-up launch: classify by the counter's size
        for tag, expect in (("full", full), ("warm-up", full / 8)):
            if 0.3 < v / expect < 3.0:
                ratios.setdefault((wr, tag), []).append(v / expect)
    for (wr, tag), rs in sorted(ratios.items()):
        print("%s  %-26s %-8s launches %3d   counter / known KiB: min %.3f  mean %.3f  max %.3f"
              % (label, "2 loads : 1 store kernels" if wr else "read-only kernels", tag, len(rs), min(rs), sum(rs) / len(rs), max(rs)))
