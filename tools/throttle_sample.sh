#!/bin/bash
# Throttle / power / clock readout while the headline bench loops ("power-limited" was inference, not evidence).
# Tries every readout this image offers; whatever works is logged verbatim.
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r04}_throttle_status_during_bench.txt; : > $O
python bench.py --no-extras --no-cpu-baseline --no-live-traffic --steps 4000 --warmup 5 > gpurun_out/${TAG:-r04}_throttle_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3; do
  echo "=== sample $i (bench loop running) $(date +%T)" >> $O
  echo "--- amd-smi metric -g 0 (power / clock / throttle / perf-level sections)" >> $O
  timeout 20 amd-smi metric -g 0 --power --clock --temperature --perf-level --throttle 2>&1 | head -120 >> $O
  echo "--- amd-smi metric -g 0 --throttle (alone)" >> $O
  timeout 20 amd-smi metric -g 0 --throttle 2>&1 | head -60 >> $O
  echo "--- rocm-smi" >> $O
  timeout 20 rocm-smi --showpower --showclocks --showtemp --showuse --showperflevel --showmaxpower 2>&1 | grep -vE "^=|^$" | head -40 >> $O
  echo "--- sysfs" >> $O
  for f in /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/power_dpm_force_performance_level /sys/class/hwmon/hwmon*/power1_average /sys/class/hwmon/hwmon*/power1_cap /sys/class/hwmon/hwmon*/freq1_input; do
    [ -r $f ] && { echo "$f:" >> $O; cat $f 2>/dev/null | head -12 >> $O; }
  done
  sleep 4
done
wait $BP
echo "=== idle, after the run" >> $O
timeout 20 amd-smi metric -g 0 --power --clock --throttle 2>&1 | head -80 >> $O
tail -c 400 gpurun_out/${TAG:-r04}_throttle_bench.json | head -c 400 >> $O
wc -l $O; grep -iE "throttl|violation|POWER|socket_power|clk|CAP" $O | head -60
