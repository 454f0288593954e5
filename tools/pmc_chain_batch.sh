#!/bin/bash
# tools/pmc_chain_batch.sh [MEMBERS] -- SQ / TA / TD / TCP / TCC counters of the batched flow chain alone (tools/chain_time.py: the 12
# launches per chain of MEMBERS pairs, 2160p HDR = 480 x 270 grid), separate --pmc passes with --kernel-trace only; the per-kernel
# reading (busy fractions, hit rates, instructions per wave) is tools/pmc_chain_derive.py.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmc_chain_batch; rm -rf $O; mkdir -p $O
M=${1:-16}
cd /tmp
pass() { n=$1; shift
  timeout 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/chain_time.py --batch $M --n 20 > $O/$n.log 2>&1; rc=$?; echo "$n rc=$rc"
  if [ $rc -ne 0 ]; then grep -m1 -i "exceeds\|error" $O/$n.log | cut -c1-200; fi; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
pass sq3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass sq4 SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass td TD_TD_BUSY_sum TD_TC_STALL_sum
pass ta TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum
pass tcp TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr
cd $R
python3 tools/pmc_chain_derive.py $O
