#!/bin/bash
for cfg in "8 2" "4 1" "8 1" "6 2" "12 3" "8 4"; do
  set -- $cfg
  for t in "" "--warp-turnstile"; do
  python bench.py --streams $1 --batch $2 $t --steps 100 --warmup 10 --no-cpu-baseline --no-reference 2>&1 | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('streams=%2d batch=%d %-16s frames/s=%8.0f  warp launch %6.1f us frac %.3f  flow %.1f us' % ($1, $2, '$t', j['value'], r['avg_launch_us'], r['frac'], 1e3*j['ms_per_flow_calc']))"
  done
done
