#!/bin/bash
# per-kernel post-processing times of library variants (yolo_v3_amd/libyv3_NAME.so built by tools/build_variant.sh): tools/r03_pp_variants.sh TAG case NAME...
export TMPDIR=/tmp
TAG=$1; CASE=$2; shift; shift
O=gpurun_out
for v in "$@"; do
  if [ $v = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so; fi
  rm -rf $O/var_prof
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/var_prof -o t -- python tools/postproc_bench.py $CASE > $O/${TAG}_$v.txt 2> $O/var_prof.err
  f=$(find $O/var_prof -name '*kernel_stats.csv' | head -1)
  echo "== $v $CASE" >> $O/${TAG}_variants.txt
  python - $f >> $O/${TAG}_variants.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "")
    if any(k in n for k in ("filter_kernel", "segpart", "subpart", "rank", "mask_kernel", "scan_kernel", "compact", "zero_kernel")):
        print("%-44s calls %4s  avg %9.1f us" % (n.split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/var_prof
done
unset YV3_LIB
cat $O/${TAG}_variants.txt
