O=gpurun_out
for rep in 1 2; do for v in "A=0" "YV3_SK=1" "YV3_SK=0"; do for lanes in 0 1; do
  env $v python bench.py --size 608 --batch 8 --weights dense --no-extras --no-cpu-baseline --no-live-traffic --lanes $lanes --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v rep$rep lanes_arg=$lanes lanes=%d  %.1f img/s  %.3f ms/step' % (d['lanes'], d['value'], d['ms_per_step']))" >> $O/pp14_config4_sk.txt
done; done; done
cat $O/pp14_config4_sk.txt
