#!/bin/bash
# tools/final_profiles.sh TAG -- the measurements the round's profiles/ are made from (run on the GPU box)
TAG=${1:-r01}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
python bench.py --workload sdr1080_24to60 --no-reference > $O/bench_sdr1080.json 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o p -- python3 $R/bench.py --no-cpu-baseline --no-reference > $O/bench_default_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_streams1 -o p -- python3 $R/bench.py --streams 1 --batch 1 --steps 40 --no-cpu-baseline --no-reference > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --streams 1 --batch 1 --steps 20 --warmup 5 --no-profile --no-cpu-baseline --no-reference > /dev/null 2>&1
  echo "pmc $c rc=$?"
done
ls $O
