#!/bin/bash
# ONE parameterised driver for everything that runs on the GPU box (replaces the per-experiment r03_*.sh / r04_*.sh one-offs).
#   gpurun --timeout 2400 -- 'bash tools/gpu.sh TAG CMD [args]; bash tools/gpu.sh TAG CMD2 ...'
# Outputs go to gpurun_out/TAG_*; copy what should be judged into profiles/ (index: profiles/README.md).
#
#   tests  [pytest args]                      pytest -m gpu (default: the whole suite)            -> TAG_tests.log
#   bench  [bench.py args]                    one bench.py run                                     -> TAG_bench.json / .err
#   benchab "LIB.." "ENV.." "bench args" [R]  same-box A/B: R (2) alternating passes over library builds (names given to
#                                             tools/build_variant.sh; `base` = libyv3.so) x environment settings (`-` = none,
#                                             e.g. "YV3_TUNE=0,8")                                 -> TAG_benchab.txt
#   tool   "LIB.." SCRIPT [ENV=V ..] -- args  python tools/SCRIPT.py args, once per library build  -> TAG_SCRIPT.log
#   prof   "bench args" B SIZE                rocprofv3 --kernel-trace --stats of a one-lane run   -> TAG_kernel_stats.csv, TAG_layers.txt
#   pmc    "bench args" DTYPE SIZE B NLAUNCH  three --pmc passes (MFMA busy / FETCH / WRITE)       -> TAG_mfma_util.json, TAG_traffic_*.json
#   plan   "B.." SIZE DTYPE                   which kernel / tile / form runs which layer, per batch size (kernel trace)  -> TAG_plan_table.txt
#   probe  NAME                               tools/probes/NAME (a hipcc-built micro-benchmark)    -> TAG_NAME.txt
export TMPDIR=/tmp
TAG=$1; CMD=$2; shift 2
# tuning / A-B environment overrides (YV3_TUNE, YV3_TILE, YV3_LIB, ...) are honoured in measurement sessions only: the A/B commands set
# YV3_MEASURE=1; `tests`, `bench`, `prof`, `pmc`, `plan` run the product as shipped with every stray YV3_* variable removed
case $CMD in benchab|tool|probe) export YV3_MEASURE=1 ;; *) for v in $(env | grep -o '^YV3_[A-Z0-9_]*' | grep -v '^YV3_DUMP_PLAN$'); do unset $v; done ;; esac
O=gpurun_out; mkdir -p $O
uselib() { if [ "$1" = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$1.so; fi; }
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%s lanes=%d  %.1f img/s  %.3f ms/step  one-lane conv %.3f ms  frac %.4f executed %.4f' % ('$1', d['config']['lanes'], d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], r.get('executed_frac', 0)))"; }
case $CMD in
tests)  [ $# -eq 0 ] && set -- tests; timeout 1800 python -m pytest "$@" -m gpu -x -q -s 2>&1 | tail -150 > $O/${TAG}_tests.log; tail -3 $O/${TAG}_tests.log ;;
bench)  timeout 900 python bench.py "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; cp $O/bench_full.json $O/${TAG}_bench_full.json 2>/dev/null; tail -c 2500 $O/${TAG}_bench.json; tail -3 $O/${TAG}_bench.err ;;
benchab) LIBS=$1; ENVS=$2; ARGS=$3; R=${4:-2}
        for rep in $(seq $R); do for l in $LIBS; do for e in $ENVS; do
          uselib $l; [ "$e" = - ] && e="YV3_NOP=1"
          env $e python bench.py $ARGS --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | line "$l $e | $ARGS | rep$rep" >> $O/${TAG}_benchab.txt
        done; done; done; unset YV3_LIB; cat $O/${TAG}_benchab.txt ;;
tool)   LIBS=$1; S=$2; shift 2; ENVV=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do ENVV+=("$1"); shift; done; shift
        for l in $LIBS; do uselib $l; echo "=== $l ${ENVV[*]} $*" >> $O/${TAG}_$S.log
          env "${ENVV[@]}" timeout 600 python tools/$S.py "$@" 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_$S.log; done
        unset YV3_LIB; cut -c1-300 $O/${TAG}_$S.log ;;
prof)   ARGS=$1; B=$2; SIZE=$3; rm -rf $O/${TAG}_prof
        YV3_DUMP_PLAN=$O/${TAG}_plan.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o t -- python bench.py $ARGS --lanes 1 --no-extras --no-cpu-baseline --no-live-traffic --steps 25 > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err
        f=$(find $O/${TAG}_prof -name '*kernel_stats.csv' | head -1); cp $f $O/${TAG}_kernel_stats.csv; head -12 $O/${TAG}_kernel_stats.csv
        t=$(find $O/${TAG}_prof -name '*kernel_trace.csv' | head -1); python tools/trace_layers.py $t $B $SIZE $O/${TAG}_plan.json > $O/${TAG}_layers.txt; tail -30 $O/${TAG}_layers.txt
        rm -rf $O/${TAG}_prof ;;
pmc)    ARGS=$1; DT=$2; SIZE=$3; B=$4; NL=$5
        for c in MFMA FETCH WRITE; do
          case $c in MFMA) ctr="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";; FETCH) ctr="FETCH_SIZE";; WRITE) ctr="WRITE_SIZE";; esac
          rm -rf $O/${TAG}_pmc_$c
          timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c -o t -- python bench.py $ARGS --lanes 1 --no-extras --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1 > /dev/null 2> $O/${TAG}_pmc_$c.err
        done
        python tools/mfma_util_summary.py $O/${TAG}_pmc_MFMA > $O/${TAG}_mfma_util.json; head -70 $O/${TAG}_mfma_util.json
        python tools/traffic_summary.py $O/${TAG}_pmc_FETCH $O/${TAG}_pmc_WRITE 4 $NL $([ "$DT" = f32 ] && echo f32r6) > $O/${TAG}_traffic_${DT}_${SIZE}_bs${B}.json; cat $O/${TAG}_traffic_${DT}_${SIZE}_bs${B}.json
        rm -rf $O/${TAG}_pmc_MFMA $O/${TAG}_pmc_FETCH $O/${TAG}_pmc_WRITE ;;
plan)   BS=$1; SIZE=$2; DT=$3; : > $O/${TAG}_plan_table.txt
        for b in $BS; do rm -rf $O/${TAG}_pl
          YV3_DUMP_PLAN=$O/${TAG}_pl.json timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_pl -o t -- python bench.py --batch $b --size $SIZE --dtype $DT --lanes 1 --no-extras --no-cpu-baseline --no-live-traffic --steps 6 --warmup 2 > /dev/null 2> $O/${TAG}_pl.err
          t=$(find $O/${TAG}_pl -name '*kernel_trace.csv' | head -1)
          echo "=== $DT ${SIZE}x${SIZE} batch $b, one lane" >> $O/${TAG}_plan_table.txt
          python tools/trace_layers.py $t $b $SIZE $O/${TAG}_pl.json >> $O/${TAG}_plan_table.txt; rm -rf $O/${TAG}_pl
        done; cat $O/${TAG}_plan_table.txt ;;
probe)  timeout 300 ./tools/probes/$1 > $O/${TAG}_$1.txt 2>&1; cat $O/${TAG}_$1.txt ;;
*) echo "unknown command $CMD"; exit 2 ;;
esac
