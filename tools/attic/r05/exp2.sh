#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r05_exp2; mkdir -p $O
python -m pytest tests/test_deferred_planes_gpu.py tests/test_batch_1080p_shapes_gpu.py tests/test_batch_period_gpu.py tests/test_fused_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
show() { python3 -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('%-26s %8.0f frames/s  ms/step %7.2f  warp-in-pipe %7.1f us  chain %6.1f us/pair' % ('$2', d['value'], d['ms_per_step'], r.get('kernel_in_pipeline',{}).get('avg_launch_us',0), 1e3*(d['ms_per_flow_calc'] or 0)))"; }
for rep in 1 2; do
for wl in sdr1080_24to60 sdr1080_64pairs; do
python bench.py $Q --workload $wl > $O/${wl}_defer_$rep.json 2>$O/err.txt; show $O/${wl}_defer_$rep.json "$wl deferred"
python bench.py $Q --workload $wl --eager-planes > $O/${wl}_eager_$rep.json 2>>$O/err.txt; show $O/${wl}_eager_$rep.json "$wl eager"
done; done
python bench.py $Q > $O/hdr.json 2>>$O/err.txt; show $O/hdr.json "hdr2160 (nt source A)"
python bench.py $Q --workload sdr1080_24to60 --timeline-out $O/tl_sdr1080.json > $O/tl_run.json 2>>$O/err.txt; show $O/tl_run.json "sdr1080 timeline"
tail -3 $O/err.txt
