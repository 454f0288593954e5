#!/bin/bash
# tools/r05/timeline.sh [TAG] -- the pipeline's timeline without a profiler (bench.py --timeline-out), bracketed by un-profiled runs of the
# same command on the same box; 2160p HDR and 1080p SDR.
export TMPDIR=/tmp; TAG=${1:-r05}; O=gpurun_out/${TAG}_tl; mkdir -p $O
Q="--steps 12 --warmup 3 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
line() { python3 -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('%-28s %8.0f frames/s  ms/step %8.3f  %s' % ('$2', d['value'], d['ms_per_step'], json.dumps(d.get('timeline'))))"; }
for wl in hdr2160_24to120 sdr1080_24to60; do
  python bench.py $Q --workload $wl > $O/${wl}_plain_a.json 2>$O/err.txt; line $O/${wl}_plain_a.json "$wl plain"
  python bench.py $Q --workload $wl --timeline-out $O/${wl}_pipeline_timeline.json > $O/${wl}_timeline_run.json 2>>$O/err.txt; line $O/${wl}_timeline_run.json "$wl timeline"
  python bench.py $Q --workload $wl > $O/${wl}_plain_b.json 2>>$O/err.txt; line $O/${wl}_plain_b.json "$wl plain"
done
tail -5 $O/err.txt
