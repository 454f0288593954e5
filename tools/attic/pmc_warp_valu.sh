#!/bin/bash
# SQ instruction counters of the warp kernels in tools/microbench.py (fused period: mode 2 and mode 0)
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_warp_valu -o p -- python3 $R/tools/microbench.py --n 10 > /dev/null 2>&1
echo rc=$?
