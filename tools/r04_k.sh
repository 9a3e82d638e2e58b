#!/bin/bash
# Round-4: full GPU suite + default bench + bf16 config A/B (round-3 selection via YV3_TUNE=0,8 vs the new per-launch choice)
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -150 > $O/r04k_tests.log; tail -3 $O/r04k_tests.log
for rep in 1 2; do for v in "YV3_TUNE=0,8" "YV3_TUNE=0,0"; do
  for cfg in "--size 608 --batch 16 --dtype bf16" "--size 416 --batch 64 --dtype bf16"; do
  env $v python bench.py $cfg --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v | $cfg | rep$rep lanes=%d  %.1f img/s  %.3f ms/step  one-lane conv %.3f ms  frac %.4f' % (d['lanes'], d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['frac']))" >> $O/r04k_bf16_selection_ab.txt
done; done; done
cat $O/r04k_bf16_selection_ab.txt
timeout 900 python bench.py > $O/r04k_bench.json 2> $O/r04k_bench.err; tail -c 600 $O/r04k_bench.json; tail -3 $O/r04k_bench.err
