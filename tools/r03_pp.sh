#!/bin/bash
# post-processing stages alone: timings + per-kernel rocprof split.  tools/r03_pp.sh TAG
export TMPDIR=/tmp
TAG=${1:-pp}; O=gpurun_out; mkdir -p $O
python tools/postproc_bench.py > $O/${TAG}_postproc_bench.txt 2> $O/${TAG}_postproc_bench.err; cat $O/${TAG}_postproc_bench.txt; tail -3 $O/${TAG}_postproc_bench.err
python tools/postproc_bench.py dense --sub 4 >> $O/${TAG}_postproc_bench.txt 2>> $O/${TAG}_postproc_bench.err; tail -2 $O/${TAG}_postproc_bench.txt
for c in dense sparse eval; do
  rm -rf $O/${TAG}_pp_prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_pp_prof -o t -- python tools/postproc_bench.py $c > /dev/null 2> $O/${TAG}_pp_prof.err
  f=$(find $O/${TAG}_pp_prof -name '*kernel_stats.csv' | head -1)
  echo "== $c" >> $O/${TAG}_postproc_kernels.txt
  python - $f >> $O/${TAG}_postproc_kernels.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "")
    if any(k in n for k in ("filter_kernel", "segpart", "subpart", "rank", "mask_kernel", "scan_kernel", "compact", "zero_kernel")):
        print("%-44s calls %4s  avg %9.1f us" % (n.split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/${TAG}_pp_prof
done
cat $O/${TAG}_postproc_kernels.txt
