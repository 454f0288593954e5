#!/bin/bash
# A-B on ONE box with an environment switch of the SAME library: ab_env.sh "VAR=VALUE" [bench args]
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_ab; mkdir -p $O
kv=$1; shift
for i in 1 2 3; do
  python bench.py --steps 8 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io "$@" > $O/a.json 2> $O/a.err
  echo -n "[base] "; python3 tools/r02/show_bench.py $O/a.json
  env $kv python bench.py --steps 8 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io "$@" > $O/b.json 2> $O/b.err
  echo -n "[$kv] "; python3 tools/r02/show_bench.py $O/b.json
done
