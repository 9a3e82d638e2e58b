#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
bash tools/gpu_round.sh r03z tests bench rehearse
ONLY_DEFAULT=1 timeout 600 python tools/latency_bench.py 1 2 4 8 16 32 > $O/r03z_latency_small_batches.txt 2>&1; grep "B=" $O/r03z_latency_small_batches.txt
bash tools/r03_prof.sh r03z "" 64 416 > $O/r03z_prof.log 2>&1; tail -28 $O/r03z_prof.log
bash tools/r03_pmc.sh r03z "" f32h2 416 64 71 > $O/r03z_pmc.log 2>&1; grep -A7 "WINO\|ping-pong" $O/r03z_mfma_util.json | head -40
