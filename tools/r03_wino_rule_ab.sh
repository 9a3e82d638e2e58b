#!/bin/bash
# same-box A/B of the Winograd selection rule: libyv3_prev.so (one lane: 0.55-1.05 rounds; two lanes: >= 0.27 rounds per launch up to 1.05) vs libyv3.so
O=gpurun_out; out=$O/r03x_wino_rule_ab.txt; : > $out
for cfg in "--batch 64" "--batch 128" "--batch 256" "--batch 96" "--batch 32"; do
for pass in 1 2; do for v in prev base; do for l in 1 2; do
  if [ $v = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so; fi
  line=$(python bench.py $cfg --lanes $l --steps 16 --warmup 5 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1)
  echo "$cfg | $v lanes=$l pass$pass $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["lanes"])')" >> $out
done; done; done; done
unset YV3_LIB; cat $out
