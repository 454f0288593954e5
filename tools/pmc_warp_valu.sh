#!/bin/bash
# tools/pmc_warp_valu.sh VARIANT... -- instructions per wave of the batched fused period warp (tools/warp_ab.py, 16-member launches) for
# several builds of the library (hopperrender_amd/lib/exp/<variant>/, "product" = the in-tree build): SQ_WAVES + SQ_INSTS_* in two --pmc
# passes with --kernel-trace only (the kernel is VALU-bound when it has the GPU to itself: instructions per wave are what to cut).
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmc_warp_valu; rm -rf $O; mkdir -p $O
cd /tmp
for v in "$@"; do
  if [ "$v" = product ]; then export HF_LIB=""; else export HF_LIB=$R/hopperrender_amd/lib/exp/$v/libhopperflow.so; fi
  timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/$v.a -o p -- python3 $R/tools/warp_ab.py --members 16 --n 6 > $O/$v.a.log 2>&1
  timeout 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/$v.b -o p -- python3 $R/tools/warp_ab.py --members 16 --n 6 > $O/$v.b.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_instr_per_wave.py $O/$v.a $O/$v.b warp_wg
done
