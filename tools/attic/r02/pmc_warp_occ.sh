#!/bin/bash
# occupancy counters of the fused period warp (tools/r02/warp_only.py <launches> <members>)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_pmc_wo; rm -rf $O; mkdir -p $O
M=${1:-16}
cd /tmp
pass() { n=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/r02/warp_only.py 3 $M > $O/$n.log 2>&1; rc=$?; echo "$n rc=$rc"
  if [ $rc -ne 0 ]; then grep -m1 -i "exceeds\|error" $O/$n.log | cut -c1-200; fi; }
pass sq2 SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass sq3 SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
pass occ MeanOccupancyPerCU MeanOccupancyPerActiveCU
pass tcp TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum
pass td TD_TD_BUSY_sum TD_TC_STALL_sum
pass ta TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr
cd $R
for p in sq2 sq3 occ tcp td ta tcc; do echo "== $p"; python3 tools/pmc_summary.py $O/$p "warp_fast_kernel" 2>&1 | tail -5; done
grep -h "warp_fast" $O/sq2/p_kernel_trace.csv | awk -F, '{print $(NF-8), $(NF-7)}' | head -3
