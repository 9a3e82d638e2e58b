#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd_conv" 2>&1 | tail -3
ARM1=1 BB=64 timeout 600 python tools/wino_ab.py c26 c13 > $O/r03_wino_ab3.log 2>&1; ARM1=1 BB=32 timeout 600 python tools/wino_ab.py c26 c13 >> $O/r03_wino_ab3.log 2>&1
ARM1=2 BB=64 timeout 600 python tools/wino_ab.py c26 c13 >> $O/r03_wino_ab3.log 2>&1; ARM1=2 BB=32 timeout 600 python tools/wino_ab.py c26 >> $O/r03_wino_ab3.log 2>&1
ARM1=1 BB=16 timeout 600 python tools/wino_ab.py c38 >> $O/r03_wino_ab3.log 2>&1
grep -v amdgpu $O/r03_wino_ab3.log
YV3_LIB=yolo_v3_amd/libyv3_tl.so timeout 300 python tools/timeline_wino.py > $O/r03_wino_gemm_timeline2.log 2>&1; grep -v amdgpu $O/r03_wino_gemm_timeline2.log
