#!/bin/bash
export TMPDIR=/tmp
bash tools/ab_bench.sh product st0 st1 st16 st18
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
for op in "48 12" "56 14" "64 16" "48 16" "36 12"; do set -- $op
python bench.py $Q --streams $1 --batch $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('op $1/$2  %8.0f frames/s' % d['value'])"
done
