#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
for op in "48 12" "56 14" "64 16" "44 11" "48 12" "64 16"; do set -- $op
python bench.py $Q --streams $1 --batch $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('op $1/$2  %8.0f frames/s' % d['value'])"
done
