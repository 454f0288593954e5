export TMPDIR=/tmp
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-clock-probe"
for rep in 1 2 3; do for v in "" "--no-profile" "--profile-every 32"; do python bench.py $Q $v 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s %8.0f frames/s' % ('$v', d['value']))"; done; done
