#!/bin/bash
# Is the first bench run on a box slower, for how long, and does an idle pause bring the state back?  (2160p HDR default workload, short runs)
Q="--no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs --no-clock-probe"
run() { python bench.py $Q "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); dv=d['device']
print('%-28s %8.0f frames/s  timed %.2f s  power %s -> %s W  copy probe %s GB/s' % ('$*', d['value'], d['timed_region_s'], dv['at_start_of_timed_region']['power_w'], dv['at_end_of_timed_region']['power_w'], dv['hbm_streams_idle_device']['streaming_copy_GBps']))"; }
run --steps 5 --warmup 0
run --steps 5 --warmup 0
run --steps 20 --warmup 5
sleep 60
run --steps 5 --warmup 0
run --steps 5 --warmup 40
sleep 180
run --steps 5 --warmup 0
run --steps 20 --warmup 5
