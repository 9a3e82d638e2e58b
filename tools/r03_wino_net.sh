#!/bin/bash
# in-network A/B of the Winograd layers under two lanes (bench main line), two alternating passes + parity tests with it on
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; out=$O/r03_wino_net_ab.log; : > $out
for pass in 1 2; do
for v in "BASE=1" "YV3_WINO=1 YV3_WINO_MIN_CIN=512" "YV3_WINO=1 YV3_WINO_MIN_CIN=256"; do
  line=$(env $v python bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1)
  echo "$v pass$pass $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["lanes"], "one-lane", d["roofline"].get("measured_with","")[:40])')" >> $out
done; done
cat $out
YV3_WINO=1 timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_configs.py -x -q -s -k "golden or config2 or hostile_whole or full_size or decision_flip or config5" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -40 > $O/r03_wino_parity.log; tail -25 $O/r03_wino_parity.log
