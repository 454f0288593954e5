#!/bin/bash
# chain-only (bench --diagnose no-warp) throughput vs pairs in flight
for sb in "4 2" "6 2" "6 3" "4 1" "3 1" "8 2" "10 2" "12 4"; do
  set -- $sb
  python bench.py --diagnose no-warp --streams $1 --batch $2 --steps 60 --warmup 10 --no-profile --no-cpu-baseline --no-reference 2>&1 | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('no-warp streams=%2d batch=%d  us/pair (prep+chain)=%6.1f' % ($1, $2, 1e3*j['ms_per_step']/$1))"
done
