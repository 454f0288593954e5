#!/bin/bash
# tools/r05/baseline.sh -- decomposition of the default pipeline on ONE box: default / eager planes / warps only / chains only,
# the chain alone (batches of 12 and 16) and the fused period warp alone.
export TMPDIR=/tmp; O=gpurun_out/r05_base; mkdir -p $O
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
show() { python3 -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %8.0f frames/s  ms/step %7.2f  warp-in-pipe %7.1f us  chain %6.1f us/pair iso-warp %s' % ('$2', d['value'], d['ms_per_step'], r.get('kernel_in_pipeline',{}).get('avg_launch_us',0), 1e3*(d['ms_per_flow_calc'] or 0), r.get('kernel_isolated',{}).get('us_per_member')))"; }
for rep in 1 2; do
python bench.py $Q > $O/default_$rep.json 2>$O/err.txt; show $O/default_$rep.json default
python bench.py $Q --eager-planes > $O/eager_$rep.json 2>>$O/err.txt; show $O/eager_$rep.json eager-planes
done
python bench.py $Q --diagnose no-flow > $O/noflow.json 2>>$O/err.txt; show $O/noflow.json no-flow
python bench.py $Q --diagnose no-warp > $O/nowarp.json 2>>$O/err.txt; show $O/nowarp.json no-warp
python bench.py $Q --workload sdr1080_24to60 > $O/sdr.json 2>>$O/err.txt; show $O/sdr.json sdr1080
python bench.py $Q --workload sdr1080_24to60 --diagnose no-flow > $O/sdr_noflow.json 2>>$O/err.txt; show $O/sdr_noflow.json sdr1080-no-flow
python bench.py $Q --workload sdr1080_24to60 --diagnose no-warp > $O/sdr_nowarp.json 2>>$O/err.txt; show $O/sdr_nowarp.json sdr1080-no-warp
python tools/chain_time.py --batch 1 12 16
python tools/chain_time.py --batch 12 16 --hdr 0 --H 1080 --W 1920
python tools/warp_ab.py --members 12 | tail -1
rocm-smi --showclocks --showpower 2>/dev/null | head -40
