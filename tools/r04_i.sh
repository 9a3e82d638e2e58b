#!/bin/bash
# Round-4 call: ordered SPLIT (8 + 4 fragment reads, builtin waits) in the ping-pong loop: timelines, layer A/B across builds, tests, bench A/B
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
echo "=== ping-pong, ordered SPLIT (default build)" > $O/r04i_wino_osplit_timeline.log
YV3_LIB=$PWD/yolo_v3_amd/libyv3_tlos.so timeout 300 python tools/timeline_wino.py 2>&1 | grep -v amdgpu.ids >> $O/r04i_wino_osplit_timeline.log
echo "=== rolling loop" >> $O/r04i_wino_osplit_timeline.log
YV3_LIB=$PWD/yolo_v3_amd/libyv3_tlroll.so timeout 300 python tools/timeline_roll.py 2>&1 | grep -v amdgpu.ids >> $O/r04i_wino_osplit_timeline.log
grep "===\|wave 0\|wave 4" $O/r04i_wino_osplit_timeline.log
for rep in 1 2; do for v in base x4 x8; do
  if [ $v = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so; fi
  echo "=== $v rep$rep" >> $O/r04i_wino_osplit_layers.log
  BB=64 timeout 300 python tools/wino_ab.py c26 c13 c52 2>&1 | grep -v amdgpu.ids >> $O/r04i_wino_osplit_layers.log
  BB=32 timeout 300 python tools/wino_ab.py c26 c13 2>&1 | grep -v amdgpu.ids >> $O/r04i_wino_osplit_layers.log
done; done
unset YV3_LIB
cut -c1-330 $O/r04i_wino_osplit_layers.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd or stream_k or conv_bn_relu or split" 2>&1 | tail -3
for rep in 1 2; do for v in base x4 x8; do
  if [ $v = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so; fi
  python bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v rep$rep lanes=%d  %.1f img/s  %.3f ms/step  one-lane conv %.3f ms' % (d['lanes'], d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step']))" >> $O/r04i_osplit_bench.txt
done; done
unset YV3_LIB; cat $O/r04i_osplit_bench.txt
