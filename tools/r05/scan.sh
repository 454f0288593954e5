#!/bin/bash
# tools/r05/scan.sh -- operating-point scan with the round-5 cache policies (streams / batch), one box
export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-clock-probe"
run() { python bench.py $Q "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %8.0f frames/s' % ('$*', d['value']))"; }
for rep in 1 2; do
for op in "48 12" "36 12" "48 16" "64 16" "32 16" "48 24" "60 12" "40 10"; do set -- $op; run --streams $1 --batch $2; done
done
for op in "36 12" "48 16" "48 12" "32 16" "64 32" "24 12"; do set -- $op; run --workload sdr1080_24to60 --streams $1 --batch $2; done
