#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
DT=bf16 BB=16 timeout 600 python tools/tile_ab.py 0,5,7 c76 c38 c19 c152 d76 p76 p38 > $O/r03_bf16_ppk_ab.log 2>&1; cat $O/r03_bf16_ppk_ab.log
DT=bf16 BB=64 timeout 600 python tools/tile_ab.py 0,5,7 c52 c26 c13 p26 >> $O/r03_bf16_ppk_ab.log 2>&1; tail -4 $O/r03_bf16_ppk_ab.log
