#!/bin/bash
# scan_op3.sh <workload> "<streams> <batch> [extra bench args]" ...
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_scan2; mkdir -p $O
wl=$1; shift
for sb in "$@"; do set -- $sb; s=$1; b=$2; shift 2
  python bench.py --workload $wl --streams $s --batch $b --steps 8 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io "$@" > $O/s.json 2> $O/s.err
  echo -n "[$*] "; python3 tools/r02/show_bench.py $O/s.json
done
