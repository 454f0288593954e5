#!/bin/bash
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_pg; mkdir -p $O
for i in 1 2; do
for f in "" "--pg-first"; do
  python bench.py --steps 8 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io $f > $O/x.json 2> $O/x.err
  echo -n "[$f] "; python3 tools/r02/show_bench.py $O/x.json; true
done; done
