"""tools/copy_profiles.py TAG -- profiles/r02_* from gpurun_out/TAG (the output of tools/final_profiles.sh TAG on the GPU box):
bench lines, rocprofv3 kernel stats, stand-alone kernel times, the warp kernel's rows of the PMC passes; then
profiles/roofline_traffic.json is regenerated from those rows (tools/pmc_traffic.py).  Nothing is typed by hand."""
import csv, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; src = os.path.join(R, "gpurun_out", tag); dst = os.path.join(R, "profiles"); rnd = "r02"
pairs = [("bench_default.json", "bench_default.json"), ("bench_sdr1080.json", "bench_sdr1080.json"),
         ("bench_default_under_rocprof.json", "bench_default_under_rocprof.json"), ("bench_sdr1080_under_rocprof.json", "bench_sdr1080_under_rocprof.json"),
         ("microbench.txt", "microbench.txt"), ("stats_default/p_kernel_stats.csv", "bench_default_kernel_stats.csv"),
         ("stats_sdr1080/p_kernel_stats.csv", "bench_sdr1080_kernel_stats.csv"), ("stats_streams1/p_kernel_stats.csv", "bench_streams1_kernel_stats.csv"),
         ("stats_chain8/p_kernel_stats.csv", "chain_batch8_kernel_stats.csv")]
for a, b in pairs:
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, f"{rnd}_{b}")); print("copied", b)
for wl in ("hdr2160_24to120", "sdr1080_24to60"):
    files = []
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = list(csv.DictReader(open(os.path.join(src, f"pmc_{wl}_{c}", "p_counter_collection.csv"))))
        keep = [r for r in rows if "warp_fast_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
        out = os.path.join(dst, f"{rnd}_warp_period_pmc_{wl}_{c}.csv")
        with open(out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
        files.append(out); print("filtered", os.path.basename(out), len(keep), "rows")
    frame_bytes = {"hdr2160_24to120": 3840 * 2160 * 3, "sdr1080_24to60": 1920 * 1080 * 3 // 2}[wl]   # P010 / NV12 output frame
    subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic.py"), wl, files[0], files[1], "--frame-bytes", str(frame_bytes)])
