#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -30 > $O/r03_wino_f32_tests.log; cat $O/r03_wino_f32_tests.log
out=$O/r03_wino_f32_net_ab.log; : > $out
for v in "YV3_WINO=0" "YV3_WINO=1" "YV3_WINO=1 YV3_LANES=1" "YV3_WINO=0 YV3_LANES=1"; do
  line=$(env $v python bench.py --dtype f32 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1)
  echo "$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["lanes"], "frac", d["roofline"]["frac"])')" >> $out
done
cat $out
YV3_WINO=1 timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_configs.py -x -q -s -k "golden or config2 or hostile_whole" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | grep -i "mode 0\|F32 \|passed\|failed\|Error" | tail -20 > $O/r03_wino_f32_parity.log; cat $O/r03_wino_f32_parity.log
bash tools/r03_prof.sh r03_f32 "--dtype f32" 64 416 | tail -28
