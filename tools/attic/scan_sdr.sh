#!/bin/bash
for sb in "8 2" "8 4" "16 4" "16 8" "32 8" "12 3"; do
  set -- $sb
  python bench.py --workload sdr1080_24to60 --streams $1 --batch $2 --steps 100 --warmup 10 --no-profile --no-cpu-baseline --no-reference 2>&1 | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('sdr1080 streams=%2d batch=%d  us/period=%7.1f  frames/s=%8.0f' % ($1, $2, 1e3*j['ms_per_step']/$1, j['value']))"
done
