#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -40 > $O/r03_wino_tests.log; cat $O/r03_wino_tests.log
BB=64 timeout 600 python tools/wino_ab.py c26 c13 c52 > $O/r03_wino_ab.log 2>&1; cat $O/r03_wino_ab.log
BB=16 timeout 600 python tools/wino_ab.py c38 c19 >> $O/r03_wino_ab.log 2>&1; tail -3 $O/r03_wino_ab.log
