#!/bin/bash
for cfg in "8 8 1" "8 8 2" "6 6 1" "4 4 1"; do
  set -- $cfg
  HF_BATCH_WARP_STREAMS=$3 python bench.py --streams $1 --batch $2 --dual-stream-contexts --steps 100 --warmup 10 --no-profile --no-cpu-baseline --no-reference 2>&1 | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('streams=%2d batch=%d warp_streams=%d  us/period=%7.1f  frames/s=%8.0f' % ($1, $2, $3, 1e3*j['ms_per_step']/$1, j['value']))"
done
