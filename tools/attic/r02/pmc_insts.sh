#!/bin/bash
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02_pmc_insts; rm -rf $O; mkdir -p $O
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -o p -- python3 $R/tools/chain_time.py --batch 8 --n 10 > /dev/null 2>&1
cd $R; python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$O/a/p_counter_collection.csv")):
    acc[r["Kernel_Name"][28:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    if "flow" not in k and "blur" not in k: continue
    w=max(d["SQ_WAVES"]); 
    sel=lambda c: max(d[c])/w
    print("%-48s waves %6d  VALU %6.0f SALU %6.0f VMEM %5.1f LDS %5.1f SMEM %5.1f per wave; wave_cycles/wave %7.0f" % (k, w, sel("SQ_INSTS_VALU"), sel("SQ_INSTS_SALU"), sel("SQ_INSTS_VMEM_RD"), sel("SQ_INSTS_LDS"), sel("SQ_INSTS_SMEM"), 4*sel("SQ_WAVE_CYCLES")))
PY
