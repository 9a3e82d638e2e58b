#!/bin/bash
# Power / clock samples (rocm-smi) while the headline bench loops: evidence for "the dominant kernel is power-limited".
O=gpurun_out/smi_sample.txt; : > $O
python bench.py --no-extras --no-cpu-baseline --no-live-traffic --steps 5000 --warmup 5 > gpurun_out/smi_bench.json 2>/dev/null &
BP=$!
n=0
for i in $(seq 1 120); do
  S=$(rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power \(W\)|sclk|mclk|junction|GPU use")
  if echo "$S" | grep -q "GPU use (%): [1-9]"; then
    n=$((n+1)); echo "--- busy sample $n" >> $O; echo "$S" >> $O
    [ $n -ge 15 ] && break
  fi
  sleep 1.5
done
wait $BP
echo "--- idle, after the run" >> $O; sleep 3
rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power \(W\)|sclk|mclk|GPU use" >> $O
rocm-smi --showmaxpower 2>/dev/null | grep -iE "Max Graphics" >> $O
