#!/bin/bash
for q in 4 5 7 8 13; do
  for s in 3 4 6 8 12; do
    GPU_MAX_HW_QUEUES=$q python bench.py --streams $s --steps 100 --warmup 10 --no-profile --no-cpu-baseline --no-reference 2>/dev/null | tail -1 | \
      python -c "import json,sys; j=json.loads(sys.stdin.read()); print('hwq=%2d streams=%2d  us/period=%7.1f  frames/s=%8.0f' % ($q, $s, 1e3*j['ms_per_step']/$s, j['value']))"
  done
done
