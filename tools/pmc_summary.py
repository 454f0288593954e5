"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean of each counter."""
import csv, glob, sys, collections
root = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if filt and filt not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
