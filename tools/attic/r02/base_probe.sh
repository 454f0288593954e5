#!/bin/bash
# round-2 baseline probe: per-kernel stats of the batched chain, default bench on today's box
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_base; mkdir -p $O
python tools/chain_time.py --batch 1 2 4 8 > $O/chain_time.txt 2>&1
python bench.py --steps 100 --no-cpu-baseline --no-reference > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_chain8 -o p -- python3 $R/tools/chain_time.py --batch 8 --n 50 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_chain1_sdr -o p -- python3 $R/tools/chain_time.py --batch 1 8 --n 50 --hdr 0 --H 1080 --W 1920 > $O/chain_time_sdr.txt 2>&1
cat $O/chain_time.txt $O/chain_time_sdr.txt; tail -c 1500 $O/bench_default.json
