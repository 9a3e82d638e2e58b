#!/bin/bash
# bf16 tile A/B + ping-pong timeline on one box
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
DT=bf16 BB=16 python tools/tile_ab.py 0,5,3 c76 c38 c19 c152 d76 d38 p76 p38 p19 > $O/r03_bf16_tile_ab.log 2>&1; cat $O/r03_bf16_tile_ab.log
DT=bf16 BB=64 python tools/tile_ab.py 0,5,3 c52 c26 c13 p52 p26 >> $O/r03_bf16_tile_ab.log 2>&1; tail -5 $O/r03_bf16_tile_ab.log
YV3_LIB=yolo_v3_amd/libyv3_tl.so DT=bf16 BB=34 python tools/timeline_pp.py > $O/r03_bf16_pp_timeline.log 2>&1; cat $O/r03_bf16_pp_timeline.log
YV3_LIB=yolo_v3_amd/libyv3_tl.so DT=f32h2 BB=64 python tools/timeline_pp.py > $O/r03_f32h2_pp_timeline.log 2>&1; tail -12 $O/r03_f32h2_pp_timeline.log
