#!/bin/bash
# Round-4 call 4: the rolling single-phase main loop (ROLL) vs the ping-pong Winograd stage and vs the bf16 four-wave tile
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for b in 64 32 16; do BB=$b timeout 300 python tools/wino_ab.py c26 c13 c52 2>&1 | grep -v amdgpu.ids >> $O/r04d_wino_roll_ab.log; done
cat $O/r04d_wino_roll_ab.log | cut -c1-330
DT=bf16 BB=16 timeout 300 python tools/tile_ab.py 0,5,7 c76 c38 c19 c152 d76 d38 p76 p38 2>&1 | grep -v amdgpu.ids >> $O/r04d_bf16_roll_ab.log
DT=bf16 BB=64 timeout 300 python tools/tile_ab.py 0,5,7 c52 c26 c13 p52 p26 2>&1 | grep -v amdgpu.ids >> $O/r04d_bf16_roll_ab.log
cat $O/r04d_bf16_roll_ab.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" 2>&1 | tail -3
