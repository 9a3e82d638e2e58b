#!/bin/bash
# bench.py at batch sizes around the two-lane threshold (detect.TWO_LANES_MIN_PIXELS = 48 images of 416x416), lanes forced to 1 and 2, alternating
SIZE=${SIZE:-416}
for b in ${@:-40 48 56}; do for l in 1 2 1 2; do
  python bench.py --size $SIZE --batch $b --lanes $l --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$SIZE bs=%d lanes=%d  %.1f img/s  %.3f ms/step' % (d['config']['global_batch'], d['config']['lanes'], d['value'], d['ms_per_step']))"
done; done
