O=gpurun_out; out=$O/r03x_wino_min_cin128_ab.txt; : > $out
for cfg in "--batch 64" "--batch 128"; do for pass in 1 2; do for v in "A=0" "YV3_WINO_MIN_CIN=128"; do for l in 1 2; do
  line=$(env $v python bench.py $cfg --lanes $l --steps 16 --warmup 5 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1)
  echo "$cfg | $v lanes=$l pass$pass $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms lanes", d["lanes"])')" >> $out
done; done; done; done
cat $out
