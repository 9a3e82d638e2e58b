#!/bin/bash
# Power / clock samples (rocm-smi) while a bench configuration loops: evidence for "the dominant kernel is power-limited".
#   bash tools/smi_sample.sh TAG "bench args" [samples]      -> gpurun_out/TAG_smi.txt (one line per busy sample + a summary line)
T=${1:-smi}; ARGS=$2; NS=${3:-10}
O=gpurun_out/${T}_smi.txt; : > $O
python bench.py $ARGS --no-extras --no-cpu-baseline --no-live-traffic --steps 100000 --warmup 5 > /dev/null 2>&1 &
BP=$!
n=0
for i in $(seq 1 150); do
  S=$(rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power \(W\)|sclk|junction|GPU use" | sed -E 's/GPU\[0\]\s*: //' | tr '\n' ';')
  if echo "$S" | grep -q "GPU use (%): [1-9]"; then n=$((n+1)); [ $n -gt 3 ] && echo "$S" >> $O; [ $n -ge $((NS+3)) ] && break; fi
  sleep 1
done
kill $BP 2>/dev/null; wait $BP 2>/dev/null
python - "$O" "$ARGS" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
p = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", t)]
c = [float(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", t)]
j = [float(x) for x in re.findall(r"junction\) \(C\): ([0-9.]+)", t)]
line = "SUMMARY bench.py %s | %d samples | package power %.0f W (min %.0f, max %.0f) | sclk %.0f MHz (min %.0f, max %.0f) | junction %.0f C" % (
    sys.argv[2] or "(headline)", len(p), sum(p) / max(len(p), 1), min(p or [0]), max(p or [0]), sum(c) / max(len(c), 1), min(c or [0]), max(c or [0]), max(j or [0]))
open(sys.argv[1], "a").write(line + "\n"); print(line)
PY
rocm-smi --showmaxpower 2>/dev/null | grep -iE "Max Graphics" | sed -E 's/GPU\[0\]\s*: //' >> $O
