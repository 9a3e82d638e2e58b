#!/bin/bash
# A/B of the Winograd lower tile bound under two lanes (YV3_OPT_TWO_LANES: 27 % instead of 55 % of a round): libyv3_rule55.so = old rule
O=gpurun_out; out=$O/r03y_wino_two_lanes_rule_ab.txt; : > $out
for cfg in "--batch 64" "--batch 32" "--batch 16" "--size 608 --batch 16" "--size 608 --batch 8 --weights dense"; do
for pass in 1 2; do for v in rule55 base; do
  if [ $v = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so; fi
  line=$(python bench.py $cfg --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1)
  echo "$cfg | $v pass$pass $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["lanes"])')" >> $out
done; done; done
unset YV3_LIB; cat $out
