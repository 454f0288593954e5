import json, sys
for p in sys.argv[1:]:
    try:
        j = json.loads(open(p).read().strip().split("\n")[-1])
    except Exception as e:
        print(p, "unreadable", e); continue
    r = j["roofline"]
    print(p.split("/")[-1], "value", j["value"], "timed_s", j.get("timed_region_s"), "host_enq_ms/step", j["host_enqueue_ms_per_step"], "ms/step", j["ms_per_step"],
          "cfg", j["config"]["pair_streams_per_gpu"], "x", j["config"]["flow_batch"], "frac", r["frac"],
          "flow", j["ms_per_flow_calc"], j["ms_per_flow_calc_isolated"],
          "warp_in", (r.get("kernel_in_pipeline") or {}).get("avg_launch_us"), "warp_iso", (r.get("kernel_isolated") or {}).get("avg_launch_us"))
