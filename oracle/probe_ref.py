"""oracle/probe_ref.py -- TEST INFRASTRUCTURE: smoke-run the compiled reference on the GPU box."""
import hashlib, os, subprocess, sys, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(os.path.dirname(here), "gpurun_out", "probe"); os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(1)
H, W = 360, 640
for i in range(3):
    f = rng.integers(0, 256, size=(H * 3 // 2, W), dtype=np.uint8)
    f.tofile(f"{out}/f{i}.bin")
script = f"""create 0 {H} {W} {W} {W} 8 6 0 255 270
radius 8
update {out}/f0.bin
update {out}/f1.bin
update {out}/f2.bin
calc
stats
dump_offsets {out}/off.bin
dump_blurred 1 {out}/blur1.bin
calc
warp 0.5 2
download {out}/o.bin
stats
time_calc 50
time_warp 50 0.5 2
"""
open(f"{out}/s.txt", "w").write(script)
r = subprocess.run([f"{here}/_ref/ref_runner", f"{out}/s.txt"], capture_output=True, text=True)
print("rc", r.returncode); print(r.stdout); print(r.stderr[-2000:])
for n in ("off.bin", "blur1.bin", "o.bin"):
    p = f"{out}/{n}"
    if os.path.exists(p): print(n, hashlib.sha256(open(p, "rb").read()).hexdigest()[:16], os.path.getsize(p))
