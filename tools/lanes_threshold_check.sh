#!/bin/bash
# bench.py at batch sizes around the two-lane threshold (detect.TWO_LANES_MIN_PIXELS = 40 images of 416x416), lanes forced to 1 and 2, alternating
#   [SIZE=608] [DTYPE=bf16|f32] bash tools/lanes_threshold_check.sh 32 36 40 48
SIZE=${SIZE:-416}; DT=${DTYPE:-f32h2}
for b in ${@:-36 40 48}; do for l in 1 2 1 2; do
  python bench.py --size $SIZE --batch $b --lanes $l --dtype $DT --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | DT=$DT SIZE=$SIZE python -c "
import json,os,sys
d=json.loads(sys.stdin.read()); print('%s %s bs=%d lanes=%d  %.1f img/s  %.3f ms/step' % (os.environ['DT'], os.environ['SIZE'], d['config']['global_batch'], d['config']['lanes'], d['value'], d['ms_per_step']))"
done; done
