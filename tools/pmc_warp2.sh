#!/bin/bash
# SQ counters of the warp kernels in tools/microbench.py for the current library and for $1 (another build)
export TMPDIR=/tmp; R=$PWD; cd /tmp
for lib in cur old; do
  [ $lib = old ] && export HF_LIB=$R/hopperrender_amd/lib/libexp_oldwarp.so
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_warp_cmp/$lib$i -o p -- python3 $R/tools/microbench.py --n 10 > /dev/null 2>&1
    echo "$lib set $i rc=$?"
  done
done
