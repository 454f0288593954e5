#!/bin/bash
for lib in "" ntp; do
  [ -n "$lib" ] && export HF_LIB=$PWD/hopperrender_amd/lib/libexp_$lib.so
  echo "lib=${lib:-default}"; python tools/microbench.py --n 20 2>&1 | grep -E 'fused period mode 2, HBM-cold|updateFrameDevice'
  for i in 1 2; do python bench.py --steps 100 --warmup 10 --no-profile --no-cpu-baseline --no-reference 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('   bench frames/s', j['value'])"; done
done
