export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-clock-probe"
for rep in 1 2; do for k in 0 1 2; do python bench.py $Q --output-memory $k 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('output-memory $k  %8.0f frames/s' % d['value'])"; done; done
for k in 0 1; do python bench.py $Q --workload sdr1080_24to60 --output-memory $k 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sdr1080 output-memory $k  %8.0f frames/s' % d['value'])"; done
