#!/bin/bash
# per-layer kernel profile of a bench configuration: tools/r03_prof.sh TAG "bench args" B SIZE
export TMPDIR=/tmp
TAG=$1; ARGS=$2; B=$3; SIZE=$4
O=gpurun_out; mkdir -p $O; rm -rf $O/${TAG}_prof
YV3_DUMP_PLAN=$O/${TAG}_plan.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o t -- python bench.py $ARGS --lanes 1 --no-extras --no-cpu-baseline --no-live-traffic --steps 25 > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err
f=$(find $O/${TAG}_prof -name '*kernel_stats.csv' | head -1); cp $f $O/${TAG}_kernel_stats.csv; head -14 $O/${TAG}_kernel_stats.csv
t=$(find $O/${TAG}_prof -name '*kernel_trace.csv' | head -1); python tools/trace_layers.py $t $B $SIZE $O/${TAG}_plan.json > $O/${TAG}_layers.txt; cat $O/${TAG}_layers.txt
rm -rf $O/${TAG}_prof
