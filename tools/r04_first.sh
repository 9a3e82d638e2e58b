#!/bin/bash
# Round-4 first GPU call: new oracle-backed tests, Winograd-loop ablation timelines, full GPU suite, default bench.
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_headline.py -x -q -s 2>&1 | tail -60 > $O/r04a_headline_tests.log; tail -5 $O/r04a_headline_tests.log
for v in 0 1 2 4 8 7; do
  echo "=== YV3_WABL=$v" >> $O/r04a_wino_ablation_timeline.log
  YV3_LIB=$PWD/yolo_v3_amd/libyv3_tlw$v.so timeout 300 python tools/timeline_wino.py 2>&1 | grep -v amdgpu.ids >> $O/r04a_wino_ablation_timeline.log
done
grep -A4 "===\|c26" $O/r04a_wino_ablation_timeline.log | grep "===\|wave 0\|wave 4" | head -40
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -150 > $O/r04a_tests.log; tail -3 $O/r04a_tests.log
timeout 900 python bench.py > $O/r04a_bench.json 2> $O/r04a_bench.err; tail -c 1500 $O/r04a_bench.json; tail -3 $O/r04a_bench.err
