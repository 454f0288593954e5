#!/bin/bash
B=${1:-8}; TAG=${2:-pmc_short}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_$TAG; rm -rf $O; mkdir -p $O
cd /tmp
pass() { n=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/chain_time.py --batch $B --n 10 > $O/$n.log 2>&1; echo "$n rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
cd $R
for p in sq1 tcp ta1 ta2 tcp2; do python3 tools/pmc_summary.py $O/$p big_partial; python3 tools/pmc_summary.py $O/$p "small_kernel<16"; done
