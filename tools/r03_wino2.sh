#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -40 > $O/r03_wino_tests2.log; cat $O/r03_wino_tests2.log
BB=64 timeout 600 python tools/wino_ab.py c26 c13 c52 > $O/r03_wino_ab2.log 2>&1; cat $O/r03_wino_ab2.log
BB=16 timeout 600 python tools/wino_ab.py c38 c19 >> $O/r03_wino_ab2.log 2>&1; tail -3 $O/r03_wino_ab2.log
BB=32 timeout 600 python tools/wino_ab.py c26 c13 >> $O/r03_wino_ab2.log 2>&1; tail -3 $O/r03_wino_ab2.log
out=$O/r03_wino_net_ab2.log; : > $out
for pass in 1 2; do
for v in "BASE=1" "YV3_WINO=1 YV3_WINO_TILE_SCHEDULE=1" "YV3_WINO=1"; do
  line=$(env $v python bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1)
  echo "$v pass$pass $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["lanes"], "one-lane", d["roofline"].get("measured_with","")[:40])')" >> $out
done; done
cat $out
