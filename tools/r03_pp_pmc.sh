#!/bin/bash
# PMC pass over the dense post-processing workload: VALU / LDS instruction counts and LDS bank conflicts per kernel
export TMPDIR=/tmp
O=gpurun_out; TAG=${1:-pp17}
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "FETCH_SIZE WRITE_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf $O/${TAG}_pmc
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${TAG}_pmc -o t -- python tools/postproc_bench.py dense > /dev/null 2> $O/${TAG}_pmc_$n.err
  f=$(find $O/${TAG}_pmc -name '*counter_collection.csv' | head -1)
  python - $f >> $O/${TAG}_postproc_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:40]
    if not any(k in n for k in ("filter_kernel", "segpart", "subpart", "rank", "mask_kernel", "scan_kernel", "compact")): continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(n, r["Counter_Name"])] += 1
for n in acc:
    print("%-40s " % n + "  ".join("%s/launch %.4g" % (c, v / cnt[(n, c)]) for c, v in sorted(acc[n].items())))
PY
  rm -rf $O/${TAG}_pmc
done
cat $O/${TAG}_postproc_pmc.txt; tail -3 $O/${TAG}_pmc_*.err | head -20
