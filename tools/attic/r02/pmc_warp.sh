#!/bin/bash
# SQ / TA counters of the fused warp kernel in tools/microbench.py
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_pmc_warp; rm -rf $O; mkdir -p $O
cd /tmp
pass() { n=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/microbench.py --n 10 > $O/$n.log 2>&1; echo "$n rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass sq2 SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
cd $R
for p in sq1 sq2 ta1 tcp; do python3 tools/pmc_summary.py $O/$p "warp_fast_kernel<unsigned short, 8, 2, 2"; done
