#!/bin/bash
# same-box A/B of BASELINE configs[4] (608x608 bs=8 SW-dense) between library builds: tools/r03_config4_ab.sh TAG NAME... (base = libyv3.so)
O=gpurun_out; TAG=$1; shift
for rep in 1 2; do for v in "$@"; do
  if [ $v = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so; fi
  for lanes in 0 1; do
    python bench.py --size 608 --batch 8 --weights dense --no-extras --no-cpu-baseline --no-live-traffic --lanes $lanes --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v rep$rep lanes_arg=$lanes lanes=%d  %.1f img/s  %.3f ms/step  stages %s' % (d['lanes'], d['value'], d['ms_per_step'], d['stages_ms']))" >> $O/${TAG}_config4_ab.txt
  done
done; done
unset YV3_LIB; cat $O/${TAG}_config4_ab.txt
