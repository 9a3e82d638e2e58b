#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "pipelined or batch_split or full_size or eval_detect or repack" -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15 > $O/r03_pipe_tests.log; cat $O/r03_pipe_tests.log
timeout 900 python bench.py > $O/r03f_bench.json 2> $O/r03f_bench.err; tail -c 300 $O/r03f_bench.json; tail -3 $O/r03f_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03f_bench.json'))
print(d['value'], d['ms_per_step'], d['lanes'])
for k,v in d.get('modes',{}).items(): print(k, v['value'], v['roofline']['frac'])
for k,v in d.get('configs',{}).items(): print(k, {kk:v[kk] for kk in v if kk in ('value','ms_per_step','gpu_ms_per_img','lanes')})
PY
bash tools/r03_pmc.sh r03p "" f32h2 416 64 71 > $O/r03p_pmc.log 2>&1; tail -40 $O/r03p_pmc.log
bash tools/r03_pmc.sh r03p_bf16 "--size 608 --batch 16 --dtype bf16" bf16 608 16 74 > $O/r03p_bf16_pmc.log 2>&1; tail -30 $O/r03p_bf16_pmc.log
bash tools/r03_pmc.sh r03p_f32 "--dtype f32" f32 416 64 74 > $O/r03p_f32_pmc.log 2>&1; tail -30 $O/r03p_f32_pmc.log
