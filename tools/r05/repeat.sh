#!/bin/bash
# tools/r05/repeat.sh -- the same short bench command several times in a row on a fresh box: does the box's state drift?
export TMPDIR=/tmp
Q="--steps 10 --no-cpu-baseline --no-reference --no-host-io --no-other-workloads"
run() { python bench.py $Q "$@" 2>/dev/null | python3 -c "
import sys,json,time
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['device']
print('%-28s %8.0f frames/s  clock %s MHz  power %s -> %s W  sclk %s  hbm copy %s fill %s  t=%d' % ('$*', d['value'], v.get('shader_clock_mhz_under_load'), v['at_start_of_timed_region'].get('power_w'), v['at_end_of_timed_region'].get('power_w'), v['at_end_of_timed_region'].get('sclk_mhz'), (v.get('hbm_streams_idle_device') or {}).get('copy_GBps_read_plus_write'), (v.get('hbm_streams_idle_device') or {}).get('fill_GBps'), time.time() % 10000))"; }
run --warmup 3
run --warmup 3
run --warmup 30
run --warmup 3
run --warmup 60
run --warmup 3
