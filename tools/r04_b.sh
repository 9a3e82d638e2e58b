#!/bin/bash
# Round-4 call 2: what bounds the Winograd GEMM stage -- memory-pattern ablations (timeline + wall), PMC of its cache traffic.
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for v in 0 16 32 48; do
  echo "=== YV3_WABL=$v" >> $O/r04b_wino_mem_ablation_timeline.log
  YV3_LIB=$PWD/yolo_v3_amd/libyv3_tlw$v.so timeout 300 python tools/timeline_wino.py 2>&1 | grep -v amdgpu.ids >> $O/r04b_wino_mem_ablation_timeline.log
done
grep "===\|wave 0\|wave 4\|chunks" $O/r04b_wino_mem_ablation_timeline.log
for v in base w16 w32; do
  if [ $v = base ]; then unset YV3_LIB; else export YV3_LIB=$PWD/yolo_v3_amd/libyv3_$v.so; fi
  echo "=== $v" >> $O/r04b_wino_mem_ablation_wall.log
  BB=64 timeout 300 python tools/wino_ab.py c26 c13 c52 2>&1 | grep -v amdgpu.ids >> $O/r04b_wino_mem_ablation_wall.log
done
unset YV3_LIB
cat $O/r04b_wino_mem_ablation_wall.log
rocprofv3 -L > $O/r04b_counters_avail.txt 2>&1; grep -c . $O/r04b_counters_avail.txt
for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum TA_BUSY_avr" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $O/pmc_tmp
  ITERS=3 BB=64 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_tmp -o t -- python tools/wino_ab.py c26 c13 > /dev/null 2> $O/r04b_pmc_$tag.err
  f=$(find $O/pmc_tmp -name '*counter_collection.csv' | head -1)
  echo "=== $set" >> $O/r04b_wino_pmc.txt
  [ -n "$f" ] && python - $f >> $O/r04b_wino_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "conv_planes_kernel" not in n and "wino_input" not in n: continue
    key = ("WINO-GEMM " if ", true>(" in n or n.rstrip(")").endswith("true>") else "direct ") + n.split("<")[1][:28] if "conv_planes" in n else "wino_input"
    key = (key, r["Grid_Size"], r["Counter_Name"])
    a = acc.setdefault(key, [0.0, 0])
    a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, g, c), (v, n) in acc.items():
    print("%-44s grid %-8s %-34s per dispatch %14.1f  (n=%d)" % (k, g, c, v / n, n))
PY
  rm -rf $O/pmc_tmp
done
cat $O/r04b_wino_pmc.txt
