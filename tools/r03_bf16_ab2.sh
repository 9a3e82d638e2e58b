#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
DT=bf16 BB=16 python tools/tile_ab.py 0,5,6 c76 c38 c19 c152 d76 d38 p76 p38 > $O/r03_bf16_tile_ab2.log 2>&1; cat $O/r03_bf16_tile_ab2.log
DT=bf16 BB=64 python tools/tile_ab.py 0,5,6 c52 c26 c13 p26 >> $O/r03_bf16_tile_ab2.log 2>&1; tail -4 $O/r03_bf16_tile_ab2.log
YV3_TILE=5 YV3_LIB=yolo_v3_amd/libyv3_tl.so DT=bf16 BB=34 python tools/timeline_np.py > $O/r03_bf16_tile5_timeline.log 2>&1; cat $O/r03_bf16_tile5_timeline.log
YV3_TILE=6 YV3_LIB=yolo_v3_amd/libyv3_tl.so DT=bf16 BB=34 python tools/timeline_np.py > $O/r03_bf16_tile6_timeline.log 2>&1; cat $O/r03_bf16_tile6_timeline.log
