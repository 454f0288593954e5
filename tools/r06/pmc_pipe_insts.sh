#!/bin/bash
# tools/r06/pmc_pipe_insts.sh WORKLOAD -- instruction counts of every kernel of the pipeline (one --pmc pass over a short bench run): the VALU / SALU /
# vector-memory instructions and waves the pipeline issues per pair and period, against the device's issue capacity at the measured frames/s
export TMPDIR=/tmp; R=$PWD; WL=${1:-sdr1080_24to60}; O=$R/gpurun_out/pmc_pipe_insts_$WL; rm -rf $O; mkdir -p $O
Q="--no-cpu-baseline --no-reference --no-host-io --no-other-workloads --no-content-legs --no-profile"
cd /tmp; timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O -o p -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --periods-per-step 8 $Q > $O/line.json 2>/dev/null
cd $R; python3 - <<PY
import csv, collections, glob, json
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("hf::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "hf::" not in r["Kernel_Name"]: continue
    acc[(k, int(r["Grid_Size"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[(k, int(r["Grid_Size"]))] += 1
d = json.loads([l for l in open("$O/line.json") if l.startswith("{")][-1]); batch = d["config"]["flow_batch"]
# the batched dispatches = the largest grid of each kernel
best = {}
for (k, g) in acc:
    if k not in best or g > best[k]: best[k] = g
tot = collections.defaultdict(float)
print("%-56s %8s %9s %9s %9s %9s" % ("kernel (batched dispatches, per pair and period)", "launches", "waves", "VALU k", "SALU k", "VMEM k"))
periods = None
for k, g in sorted(best.items()):
    a = acc[(k, g)]; L = n[(k, g)]
    if "warp" in k: periods = L
for k, g in sorted(best.items()):
    a = acc[(k, g)]; L = n[(k, g)]
    per = 1.0 / (periods * batch)
    print("%-56s %8.2f %9.0f %9.1f %9.1f %9.1f" % (k[:56], L / periods, a["SQ_WAVES"] * per, a["SQ_INSTS_VALU"] * per / 1e3, a["SQ_INSTS_SALU"] * per / 1e3, a["SQ_INSTS_VMEM_RD"] * per / 1e3))
    for c in a: tot[c] += a[c] * per
print("total per pair and period: waves %.0f  VALU %.2f M  SALU %.2f M  VMEM_RD %.1f k wave instructions" % (tot["SQ_WAVES"], tot["SQ_INSTS_VALU"] / 1e6, tot["SQ_INSTS_SALU"] / 1e6, tot["SQ_INSTS_VMEM_RD"] / 1e3))
PY
