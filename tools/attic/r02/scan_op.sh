#!/bin/bash
# operating-point scan of bench.py: "streams batch [extra flags]" per line
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_scan; mkdir -p $O
W=${WORKLOAD:-hdr2160_24to120}
i=0
while read -r s b extra; do
  [ -z "$s" ] && continue
  i=$((i+1))
  python bench.py --workload $W --streams $s --batch $b --steps 6 --warmup 2 --periods-per-step 32 --no-cpu-baseline --no-reference --no-host-io $extra > $O/b$i.json 2> $O/b$i.err
  echo -n "[$s x $b $extra] "; python3 tools/r02/show_bench.py $O/b$i.json
done
