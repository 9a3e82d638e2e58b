bash tools/gpu.sh r04m tests
bash tools/gpu.sh r04m bench
bash tools/gpu.sh r04m prof "" 64 416
bash tools/gpu.sh r04m pmc "" f32h2 416 64 71
bash tools/gpu.sh r04m plan "1 16 32 64" 416 f32h2
bash tools/gpu.sh r04m_bf16 prof "--size 608 --batch 16 --dtype bf16" 16 608
bash tools/gpu.sh r04m_bf16 pmc "--size 608 --batch 16 --dtype bf16" bf16 608 16 71
YV3_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --lanes 1 > gpurun_out/r04m_bench_2rank_gloo_rehearsal.json 2> gpurun_out/r04m_rehearse2.err; tail -c 900 gpurun_out/r04m_bench_2rank_gloo_rehearsal.json
YV3_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --global-batch 256 --steps 5 --warmup 2 --lanes 1 --no-extras > gpurun_out/r04m_bench_8rank_gloo_rehearsal_config3.json 2> gpurun_out/r04m_rehearse8.err; tail -c 1500 gpurun_out/r04m_bench_8rank_gloo_rehearsal_config3.json
