#!/bin/bash
# tools/scan_streams.sh -- diagnostic scan: per-period time vs number of pair streams, chain only / warps only / both
for d in no-warp no-flow ""; do
  for s in 1 2 4 6 8 12; do
    python bench.py --streams $s --steps 100 --warmup 10 --no-profile --no-cpu-baseline --no-reference ${d:+--diagnose $d} 2>/dev/null | tail -1 | \
      python -c "import json,sys; j=json.loads(sys.stdin.read()); print('diag=%-8s streams=%2d  us/period=%7.1f  frames/s=%8.0f' % ('$d' or 'full', $s, 1e3*j['ms_per_step']/$s, j['value']))"
  done
done
