#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; out=$O/r03_bf16_lanes_tile_threshold_ab.log; : > $out
for pass in 1 2; do
for v in "YV3_LANES=2" "YV3_LANES=2 YV3_TUNE=0,0,160" "YV3_LANES=2 YV3_TUNE=0,0,90" "YV3_LANES=1" "YV3_LANES=1 YV3_TUNE=0,0,160"; do
  line=$(env $v python bench.py --size 608 --batch 16 --dtype bf16 --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1)
  echo "$v pass$pass $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["lanes"])')" >> $out
done; done
cat $out
