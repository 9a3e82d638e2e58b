O=gpurun_out; out=$O/r03y_lane_choice_order.txt; : > $out
for cfg in "--batch 16" "--size 608 --batch 8 --weights dense"; do
for l in 0 1 0 1 1 0 1 0; do
  line=$(python bench.py $cfg --lanes $l --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1)
  echo "$cfg | --lanes $l $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["lanes"], d.get("lanes_calibration_ms"))')" >> $out
done; done
cat $out
