#!/bin/bash
# PMC passes over the batched flow chain (tools/chain_time.py --batch B): where do the waves spend their cycles
B=${1:-8}; TAG=${2:-pmc_chain$B}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_$TAG; mkdir -p $O
cd /tmp
[ -f $R/gpurun_out/rocprof_counters.txt ] || rocprofv3 -L > $R/gpurun_out/rocprof_counters.txt 2>&1
pass() { n=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/chain_time.py --batch $B --n 20 > $O/$n.log 2>&1; echo "$n rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass sq2 SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
pass ta TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass fetch FETCH_SIZE
ls $O
