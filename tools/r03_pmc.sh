#!/bin/bash
# PMC passes (MFMA busy / clock, FETCH_SIZE, WRITE_SIZE: each in its own run, kernel-trace only) of one bench configuration:
#   tools/r03_pmc.sh TAG "bench args" DTYPE SIZE B CONV_LAUNCHES_PER_STEP
export TMPDIR=/tmp
TAG=$1; ARGS=$2; DT=$3; SIZE=$4; B=$5; NL=$6
O=gpurun_out; mkdir -p $O
for c in MFMA FETCH WRITE; do
  case $c in MFMA) ctr="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";; FETCH) ctr="FETCH_SIZE";; WRITE) ctr="WRITE_SIZE";; esac
  rm -rf $O/${TAG}_pmc_$c
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c -o t -- python bench.py $ARGS --lanes 1 --no-extras --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1 > /dev/null 2> $O/${TAG}_pmc_$c.err
done
python tools/mfma_util_summary.py $O/${TAG}_pmc_MFMA > $O/${TAG}_mfma_util.json; head -70 $O/${TAG}_mfma_util.json
python tools/traffic_summary.py $O/${TAG}_pmc_FETCH $O/${TAG}_pmc_WRITE 4 $NL > $O/${TAG}_traffic_${DT}_${SIZE}_bs${B}.json; cat $O/${TAG}_traffic_${DT}_${SIZE}_bs${B}.json
rm -rf $O/${TAG}_pmc_MFMA $O/${TAG}_pmc_FETCH $O/${TAG}_pmc_WRITE
