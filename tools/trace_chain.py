"""Print per-kernel durations of the last flow chain from a rocprofv3 kernel trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].replace("void hf::(anonymous namespace)::", "").replace("hf::(anonymous namespace)::", "")[:28],
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
idx = [i for i, s in enumerate(seq) if "blur_flow" in s[0]]
i1, i2 = idx[-2] + 1, idx[-1] + 1
t0 = seq[i1][2]
out = []
for s in seq[i1:i2]:
    out.append(f"{s[0]}:{s[1]:.1f}@{(s[2]-t0)/1e3:.0f}")
print(" | ".join(out))
print("chain span us", (seq[i2 - 1][3] - t0) / 1e3)
