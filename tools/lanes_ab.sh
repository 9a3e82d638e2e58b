#!/bin/bash
# Same-box A/B of the kernel-selection switches UNDER TWO LANES (bench.py main line only, 2 alternating passes).
# usage: bash tools/lanes_ab.sh OUT.log
out=${1:-gpurun_out/lanes_ab.log}; : > $out
for pass in 1 2; do
for v in "BASE=1" "YV3_TILE=3" "YV3_TILE=2" "YV3_TUNE=0,1" "YV3_BIG_MIN=32" "YV3_BIG_MIN=300" "YV3_SK=1" "YV3_NO_FUSED_RES64=1" "YV3_NO_FUSED_FRONT=1" "YV3_LANES=1"; do
  line=$(env $v python bench.py --steps 40 --warmup 8 --no-extras 2>/dev/null | tail -1)
  echo "$v pass$pass $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["config"].get("lanes"))')" >> $out
done; done
cat $out
