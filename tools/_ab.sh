export TMPDIR=/tmp DT=f32h2 ITERS=3 CHECK=0
O=gpurun_out/pmcX; rm -rf $O; mkdir -p $O
i=0
for ctr in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/p$i -o t -- python tools/conv_bench.py d208 c208 p208 c104 p52 c52 > /dev/null 2> $O/p$i.err
  f=$(find $O/p$i -name '*counter_collection.csv' | head -1); d=$(dirname $f); pre=$(basename $f _counter_collection.csv)
  echo "=== pass $i: $ctr"; python tools/pmc_summary.py $d $pre 2>&1 | tail -20
  rm -rf $O/p$i
done
