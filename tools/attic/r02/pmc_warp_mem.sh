#!/bin/bash
# memory-path counters of the fused period warp (tools/r02/warp_only.py); few counters per block and pass (more "exceed the hardware")
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_pmc_wm; rm -rf $O; mkdir -p $O
cd /tmp
pass() { n=$1; shift
  timeout 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/r02/warp_only.py 4 > $O/$n.log 2>&1; rc=$?; echo "$n rc=$rc"
  if [ $rc -ne 0 ]; then grep -m1 -i "exceeds\|error" $O/$n.log | cut -c1-200; fi; }
pass td TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
pass ta1 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum
pass tcp1 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp3 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pass tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr
pass tcc2 TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum
pass tcc3 TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum
cd $R
for p in td ta1 ta2 tcp1 tcp2 tcp3 tcc1 tcc2 tcc3; do echo "== $p"; python3 tools/pmc_summary.py $O/$p "warp_fast_kernel" 2>&1 | tail -6; done
