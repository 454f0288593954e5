#!/bin/bash
# tools/r05/scan1080.sh -- operating point of the 1080p SDR workload (pair streams / pairs per batch), two alternating repetitions
export TMPDIR=/tmp
Q="--workload sdr1080_24to60 --steps 16 --warmup 3 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
for rep in 1 2; do for op in "36 12" "48 16" "48 12" "32 16" "54 18" "72 24" "60 20" "36 12"; do set -- $op
python bench.py $Q --streams $1 --batch $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('op $1/$2  %8.0f frames/s  chain %.1f us/pair' % (d['value'], 1e3*(d['ms_per_flow_calc'] or 0)))"
done; done
