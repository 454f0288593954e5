#!/bin/bash
# tools/pmc_warp.sh -- SQ counters of the fused period warp, HBM-cold (bench --diagnose no-flow, one stream)
export TMPDIR=/tmp; R=$PWD; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_warp_sq/$i -o p -- python3 $R/bench.py --diagnose no-flow --streams 1 --steps 20 --warmup 5 --no-profile --no-cpu-baseline --no-reference > /dev/null 2>&1
  echo "set $i rc=$?"
done
