export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-clock-probe"
for rep in 1 2; do for wl in hdr2160_24to120 hdr2160_24to60; do python bench.py $Q --workload $wl 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; pp=c['source_periods_per_step']*d['steps']
print('%-18s %8.0f frames/s  %.2f us per pair and period  outputs per period %.3f' % ('$wl', d['value'], 1e3*d['ms_per_step']*d['steps']/pp, c['output_frames_total']/pp))"; done; done
