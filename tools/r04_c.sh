#!/bin/bash
# Round-4 call 3: schedule experiments of the ping-pong loop (setprio around the MFMA cluster, SPLIT on/off): timelines, layer A/B, bench A/B
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for v in tlw0 tlx1 tlx2 tlx3 tlw3; do
  echo "=== $v" >> $O/r04c_pp_schedule_timeline.log
  YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so timeout 300 python tools/timeline_wino.py 2>&1 | grep -v amdgpu.ids >> $O/r04c_pp_schedule_timeline.log
done
grep "===\|wave 0\|wave 4" $O/r04c_pp_schedule_timeline.log
for rep in 1 2; do for v in base x1 x2 x3; do
  if [ $v = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so; fi
  echo "=== $v rep$rep" >> $O/r04c_pp_schedule_layers.log
  BB=64 timeout 300 python tools/wino_ab.py c26 c13 c52 2>&1 | grep -v amdgpu.ids >> $O/r04c_pp_schedule_layers.log
  BB=32 timeout 300 python tools/wino_ab.py c26 c13 2>&1 | grep -v amdgpu.ids >> $O/r04c_pp_schedule_layers.log
done; done
unset YV3_LIB
cat $O/r04c_pp_schedule_layers.log | cut -c1-230
for rep in 1 2; do for v in base x1 x2 x3; do
  if [ $v = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so; fi
  python bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v rep$rep lanes=%d  %.1f img/s  %.3f ms/step  one-lane conv %.3f ms' % (d['lanes'], d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step']))" >> $O/r04c_pp_schedule_bench.txt
done; done
unset YV3_LIB; cat $O/r04c_pp_schedule_bench.txt
