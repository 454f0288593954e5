#!/bin/bash
# memory-path counters of the batched fused warp (16 members per launch, warp-only pipeline: --diagnose no-flow); optional HF_LIB
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_pmc_wb; rm -rf $O; mkdir -p $O
cd /tmp
pass() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/bench.py --steps 1 --warmup 1 --periods-per-step 6 --no-profile --no-cpu-baseline --no-reference --no-host-io --streams 16 --batch 16 --diagnose no-flow > $O/$n.log 2>&1; echo "$n rc=$?"; }
pass ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_TA_BUSY_sum
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
pass tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
pass tcc2 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum
pass sq GRBM_GUI_ACTIVE SQ_BUSY_CYCLES TD_TD_BUSY_sum TD_TC_STALL_sum SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass tcc3 TCC_CYCLE_sum TCC_BUSY_sum TCC_IB_STALL_sum TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum TCC_RW_REQ_sum
cd $R
for p in ta tcp1 tcp2 tcc1 tcc2 sq tcc3; do echo "== $p"; python3 tools/pmc_summary.py $O/$p "warp_fast_kernel"; done
