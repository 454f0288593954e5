#!/bin/bash
# memory-path counters of the fused warp in tools/microbench.py (single-stream: the multi-stream bench does not finish under these counters)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_pmc_wm; rm -rf $O; mkdir -p $O
cd /tmp
pass() { n=$1; shift
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/microbench.py --n 6 > $O/$n.log 2>&1; echo "$n rc=$?"; }
pass td TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE
pass ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
pass tcp1 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
pass tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
pass tcc2 TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_CYCLE_sum TCC_BUSY_sum
cd $R
for p in td ta tcp1 tcc1 tcc2; do echo "== $p"; python3 tools/pmc_summary.py $O/$p "warp_fast_kernel<unsigned short, 8, 2, 2" 2>&1 | tail -12; done
