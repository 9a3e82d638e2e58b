#!/bin/bash
# The round's evidence run on one MI355X box: bash tools/round_final.sh TAG   (outputs gpurun_out/TAG_*; see profiles/README.md)
# bench.py's default (= the headline) is the exact-fp32 mode since round 6; the other modes are profiled by name.
T=${1:-r06zz}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
bash tools/gpu.sh $T tests
bash tools/gpu.sh $T bench
bash tools/gpu.sh $T prof "--dtype f32" 64 416
bash tools/gpu.sh $T pmc "--dtype f32" f32 416 64 71
bash tools/gpu.sh $T plan "1 8 16 32 64" 416 f32
bash tools/gpu.sh ${T}_f32h2 prof "--dtype f32h2" 64 416
bash tools/gpu.sh ${T}_f32h2 pmc "--dtype f32h2" f32h2 416 64 71
bash tools/gpu.sh ${T}_bf16 prof "--size 608 --batch 16 --dtype bf16" 16 608
YV3_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --lanes 1 > gpurun_out/${T}_bench_2rank_gloo_rehearsal.json 2> gpurun_out/${T}_rehearse2.err; tail -c 600 gpurun_out/${T}_bench_2rank_gloo_rehearsal.json
YV3_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --global-batch 256 --steps 5 --warmup 2 --lanes 1 --no-extras > gpurun_out/${T}_bench_8rank_gloo_rehearsal_config3.json 2> gpurun_out/${T}_rehearse8.err; tail -c 600 gpurun_out/${T}_bench_8rank_gloo_rehearsal_config3.json
