"""tools/copy_profiles.py TAG -- profiles/TAG_* from gpurun_out/TAG (the output of tools/final_profiles.sh TAG on the GPU box):
bench lines, rocprofv3 kernel stats, stand-alone kernel times, the PMC passes reduced to one row per kernel and launch shape;
then profiles/roofline_traffic.json is regenerated from those rows (tools/pmc_traffic.py: the dominant kernel's bytes per
launch AND the whole pipeline's bytes per output frame).  Nothing is typed by hand."""
import csv, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; src = os.path.join(R, "gpurun_out", tag); dst = os.path.join(R, "profiles"); rnd = tag
pairs = [("bench_default.json", "bench_default.json"), ("bench_sdr1080.json", "bench_sdr1080.json"),
         ("bench_sdr1080_64pairs.json", "bench_sdr1080_64pairs.json"), ("bench_hdr2160_nb10_blur32.json", "bench_hdr2160_nb10_blur32.json"),
         ("bench_default_under_rocprof.json", "bench_default_under_rocprof.json"), ("bench_sdr1080_under_rocprof.json", "bench_sdr1080_under_rocprof.json"),
         ("microbench.txt", "microbench.txt"), ("stats_default/p_kernel_stats.csv", "bench_default_kernel_stats.csv"),
         ("stats_sdr1080/p_kernel_stats.csv", "bench_sdr1080_kernel_stats.csv"), ("stats_streams1/p_kernel_stats.csv", "bench_streams1_kernel_stats.csv"),
         ("stats_chain16/p_kernel_stats.csv", "chain_batch16_kernel_stats.csv"), ("pmc_warp_valu.txt", "pmc_warp_instructions.txt"),
         ("pmc_warp_wg_kernel.txt", "pmc_warp_wg_kernel.txt"), ("pmc_chain_batch16.txt", "pmc_chain_batch16.txt"), ("chain_beside_warp.txt", "chain_beside_warp.txt"),
         ("timeline_vs_trace_hdr2160_24to120.txt", "timeline_vs_trace_hdr2160_24to120.txt"), ("timeline_vs_trace_sdr1080_24to60.txt", "timeline_vs_trace_sdr1080_24to60.txt"),
         ("pipeline_timeline.json", "pipeline_timeline.json"), ("pipeline_timeline_sdr1080.json", "pipeline_timeline_sdr1080.json"),
         ("bench_default_timeline_run.json", "bench_default_timeline_run.json"), ("bench_default_plain_after_timeline.json", "bench_default_plain_after_timeline.json"),
         ("bench_hdr1080.json", "bench_hdr1080.json"), ("bench_sdr2160.json", "bench_sdr2160.json"), ("bench_default_wrap6.json", "bench_default_wrap6.json"),
         ("bench_default_no_sad_reuse.json", "bench_default_no_sad_reuse.json"), ("bench_sdr1080_no_sad_reuse.json", "bench_sdr1080_no_sad_reuse.json"),
         ("stats_chain16_static/p_kernel_stats.csv", "chain_batch16_static_kernel_stats.csv"), ("stats_chain16_chaotic/p_kernel_stats.csv", "chain_batch16_chaotic_kernel_stats.csv"),
         ("stats_chain16_noreuse/p_kernel_stats.csv", "chain_batch16_no_sad_reuse_kernel_stats.csv")]
for a, b in pairs:
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, f"{rnd}_{b}")); print("copied", b)
    else:
        print("MISSING", a)
frame_bytes = {"hdr2160_24to120": 3840 * 2160 * 3, "sdr1080_24to60": 1920 * 1080 * 3 // 2,   # P010 / NV12 output frame
               "hdr1080_24to120": 1920 * 1080 * 3, "sdr2160_24to60": 3840 * 2160 * 3 // 2}
outputs_per_period = {"hdr2160_24to120": 417083 / 83333, "sdr1080_24to60": 417083 / 166667,     # source / target frame time (HopperRender.cpp:162-163)
                      "hdr1080_24to120": 417083 / 83333, "sdr2160_24to60": 417083 / 166667}


# pairs per batch at the bench's operating point (the PMC passes over the pipeline run the same default command)
import json
flow_batch = {}
for wl, f in (("hdr2160_24to120", "bench_default.json"), ("sdr1080_24to60", "bench_sdr1080.json"), ("hdr1080_24to120", "bench_hdr1080.json"), ("sdr2160_24to60", "bench_sdr2160.json")):
    try:
        flow_batch[wl] = json.loads([l for l in open(os.path.join(src, f)) if l.startswith("{")][-1])["config"]["flow_batch"]
    except Exception:
        flow_batch[wl] = 16


def reduce_rows(path, counter, keep):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and keep(r["Kernel_Name"])]
    return rows


# `python tools/copy_profiles.py TAG lines`: only the summaries above (after `final_profiles.sh TAG bench`): the PMC rows and
# profiles/roofline_traffic.json -- whose commit hash the bench lines print -- stay as the full run left them
for wl in (() if sys.argv[2:3] == ["lines"] else ("hdr2160_24to120", "sdr1080_24to60", "hdr1080_24to120", "sdr2160_24to60")):
    for kind, prefix, keep, extra in (("warp_period", "pmc", lambda n: "::warp_" in n, []),
                                      ("pipeline", "pmcpipe", lambda n: "hf::" in n, ["--pipeline", "--batch", str(flow_batch[wl]), "--outputs-per-period", "%.5f" % outputs_per_period[wl]])):
        if kind == "warp_period" and wl in ("hdr1080_24to120", "sdr2160_24to60"):
            continue       # (round 6 added these two with pipeline passes only)
        files = []
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            p = os.path.join(src, f"{prefix}_{wl}_{c}", "p_counter_collection.csv")
            if not os.path.exists(p):
                print("MISSING", p); files = []; break
            rows = reduce_rows(p, c, keep)
            out = os.path.join(dst, f"{rnd}_{kind}_pmc_{wl}_{c}.csv")
            cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"]
            with open(out, "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=cols, extrasaction="ignore"); w.writeheader(); w.writerows(rows)
            files.append(out); print("reduced", os.path.basename(out), len(rows), "rows")
        if files:
            subprocess.check_call([sys.executable, os.path.join(R, "tools", "pmc_traffic.py"), wl, files[0], files[1], "--frame-bytes", str(frame_bytes[wl])] + extra)
