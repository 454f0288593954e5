#!/bin/bash
for q in 5; do
for sb in "4 1" "8 2" "12 3" "16 4" "8 1" "4 2" "6 2"; do
  set -- $sb
  GPU_MAX_HW_QUEUES=$q python bench.py --streams $1 --batch $2 --steps 100 --warmup 10 --no-profile --no-cpu-baseline --no-reference 2>&1 | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('hwq=%d streams=%2d batch=%d  us/period=%7.1f  frames/s=%8.0f' % ($q, $1, $2, 1e3*j['ms_per_step']/$1, j['value']))"
done
done
