#!/bin/bash
# does the lane calibration pick the faster schedule?  per config: forced 1 lane, forced 2 lanes, automatic (x2)
O=gpurun_out; out=$O/r03y_lane_choice.txt; : > $out
for cfg in "--batch 64" "--batch 32" "--batch 16" "--size 608 --batch 16 --dtype bf16" "--size 608 --batch 8 --weights dense"; do
for l in 1 2 0 0; do
  line=$(python bench.py $cfg --lanes $l --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1)
  echo "$cfg | --lanes $l $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["lanes"], d.get("lanes_calibration_ms"))')" >> $out
done; done
cat $out
