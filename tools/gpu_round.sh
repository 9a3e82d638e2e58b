#!/bin/bash
# One GPU session: tests, bench line, rocprofv3 kernel stats, PMC passes (MFMA util, FETCH, WRITE).  Usage:
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r02a [tests] [bench] [prof] [pmc] [rehearse]'
TAG=$1; shift
WHAT="${@:-tests bench prof pmc}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for w in $WHAT; do case $w in
tests) timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -150 > $O/${TAG}_tests.log; tail -3 $O/${TAG}_tests.log ;;
bench) timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -c 600 $O/${TAG}_bench.json; tail -3 $O/${TAG}_bench.err ;;
prof)  rm -rf $O/${TAG}_prof; YV3_DUMP_PLAN=$O/${TAG}_plan.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o t -- python bench.py --lanes 1 --no-extras --no-cpu-baseline --no-live-traffic --steps 25 > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err
       f=$(find $O/${TAG}_prof -name '*kernel_stats.csv' | head -1); cp $f $O/${TAG}_kernel_stats.csv; head -12 $O/${TAG}_kernel_stats.csv
       t=$(find $O/${TAG}_prof -name '*kernel_trace.csv' | head -1); python tools/trace_layers.py $t 64 416 $O/${TAG}_plan.json > $O/${TAG}_layers.txt; tail -30 $O/${TAG}_layers.txt
       cp $t $O/${TAG}_kernel_trace.csv 2>/dev/null; rm -rf $O/${TAG}_prof ;;
pmc)   for c in MFMA FETCH WRITE; do
         case $c in MFMA) ctr="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";; FETCH) ctr="FETCH_SIZE";; WRITE) ctr="WRITE_SIZE";; esac
         rm -rf $O/${TAG}_pmc_$c
         timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c -o t -- python bench.py --lanes 1 --no-extras --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1 > /dev/null 2> $O/${TAG}_pmc_$c.err
       done
       python tools/mfma_util_summary.py $O/${TAG}_pmc_MFMA > $O/${TAG}_mfma_util.json; cat $O/${TAG}_mfma_util.json | head -60
       python tools/traffic_summary.py $O/${TAG}_pmc_FETCH $O/${TAG}_pmc_WRITE 4 71 > $O/${TAG}_traffic_f32h2_416_bs64.json; cat $O/${TAG}_traffic_f32h2_416_bs64.json
       rm -rf $O/${TAG}_pmc_MFMA $O/${TAG}_pmc_FETCH $O/${TAG}_pmc_WRITE ;;
rehearse) # bench.py's OWN N-rank launch (python bench.py --gpus 2 re-launches itself under torch.distributed.run) with 2 ranks on this ONE GPU
       # (gloo carries the gather: RCCL refuses two ranks per device).  --lanes 1: two PROCESSES sharing a GPU oversubscribe its hardware
       # queues once each also runs two lanes; one process per GPU -- the real launch -- is what lanes are calibrated for.
       # First the failure mode: without the override a 2-GPU request on a 1-GPU box must be an error, not a 1-rank run.
       python bench.py --gpus 2 --no-extras > $O/${TAG}_bench_2gpus_on_1gpu_box.out 2> $O/${TAG}_bench_2gpus_on_1gpu_box.err; echo "rc=$?" >> $O/${TAG}_bench_2gpus_on_1gpu_box.err; tail -2 $O/${TAG}_bench_2gpus_on_1gpu_box.err
       YV3_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --lanes 1 > $O/${TAG}_bench_2rank_gloo_rehearsal.json 2> $O/${TAG}_rehearse.err; tail -c 600 $O/${TAG}_bench_2rank_gloo_rehearsal.json; tail -2 $O/${TAG}_rehearse.err ;;
esac; done
