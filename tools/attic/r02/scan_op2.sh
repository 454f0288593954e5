#!/bin/bash
# operating points (streams x batch) for a workload: scan_op2.sh <workload> "s b" "s b" ...
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_scan2; mkdir -p $O
wl=$1; shift
for sb in "$@"; do set -- $sb
  python bench.py --workload $wl --streams $1 --batch $2 --steps 8 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io > $O/s.json 2> $O/s.err
  python3 tools/r02/show_bench.py $O/s.json
done
