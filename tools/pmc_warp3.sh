#!/bin/bash
export TMPDIR=/tmp; R=$PWD; cd /tmp
export HF_WARP_SPLIT=4 HF_WARP_UPW=4
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_warp_cmp2/cur$i -o p -- python3 $R/tools/microbench.py --n 10 > /dev/null 2>&1
  echo "set $i rc=$?"
done
