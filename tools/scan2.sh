#!/bin/bash
for lib in "" $PWD/hopperrender_amd/lib/libexp_xonly.so; do
for d in no-warp ""; do
  for s in 1 6; do
    HF_LIB=$lib python bench.py --streams $s --steps 100 --warmup 10 --no-profile --no-cpu-baseline --no-reference ${d:+--diagnose $d} 2>/dev/null | tail -1 | \
      python -c "import json,sys; j=json.loads(sys.stdin.read()); print('lib=%-6s diag=%-8s streams=%2d  us/period=%7.1f  frames/s=%8.0f' % ('xonly' if '$lib' else 'std', '$d' or 'full', $s, 1e3*j['ms_per_step']/$s, j['value']))"
  done
done
done
