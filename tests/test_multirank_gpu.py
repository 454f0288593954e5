"""GPU: bench.py's multi-rank path (one process per rank under torch.distributed.run, barrier + MAX/SUM reductions of
the timing, pair streams sharded with no data-path collective) executed for real.  The box has ONE GPU, so the two ranks
share it and the reductions run over gloo (HF_BENCH_BACKEND=gloo); on an 8-GPU node the driver uses nccl (= RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_bench(extra, world, workload="sdr1080_24to60", host_io=False):
    env = dict(os.environ, HF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--steps", "2", "--warmup", "1", "--periods-per-step", "4", "--workload", workload, "--no-cpu-baseline",
            "--no-reference"] + ([] if host_io else ["--no-host-io"]) + extra
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu():
    one = run_bench([], 1)
    two = run_bench([], 2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["output_frames_total"] == 2 * one["config"]["output_frames_total"]   # SUM over ranks; weak scaling
    assert two["value"] > 0 and two["scaling"] == "weak" and two["config"]["parallelism"].startswith("pair-sharded x2")
    assert "cpu_baseline" not in two and "host_io" not in two                                  # rank-0-at-N=1 only
    assert two["roofline"]["bound"] == "hbm" and 0 < two["roofline"]["frac"] < 1
    # the ranks shared one GPU: per-rank rate about halves, the aggregate stays in the same range
    assert two["value"] > 0.4 * one["value"]


def test_named_baseline_configs_4_and_5():
    """BASELINE.json configs 4 and 5 by name: `sdr1080_64pairs` (64 independent 1080p SDR pairs in flight, split over the
    ranks: strong scaling) and `hdr2160_nb10_blur32` (neighbour scalar 10, blurFlow radius 32)."""
    one = run_bench([], 1, "sdr1080_64pairs")
    assert one["config"]["pair_streams_per_gpu"] == 64 and one["config"]["flow_batch"] == 16 and one["scaling"] == "strong"
    two = run_bench([], 2, "sdr1080_64pairs")
    assert two["config"]["pair_streams_per_gpu"] == 32 and two["config"]["pair_streams_total"] == 64
    assert two["config"]["output_frames_total"] == one["config"]["output_frames_total"]        # the job is the same 64 pairs
    five = run_bench([], 1, "hdr2160_nb10_blur32")
    assert five["config"]["neighbor_scalar"] == 10 and five["config"]["blur_radius"] == 32 and five["value"] > 0
    assert five["config"]["host_calls_per_batch_and_period"] == 1
    assert 0 < five["roofline"]["frac"] < 1 and five["roofline"]["frac_algorithmic"] > five["roofline"]["frac"]


def test_host_io_leg_at_two_ranks():
    """bench.py at N > 1: every rank runs the asynchronous host-I/O path (pinned buffers, H2D / D2H on side streams) on its
    GPU at the same time and rank 0 reports the aggregate -- the leg SURVEY.md 8(e) names as the expected scaling limit."""
    two = run_bench([], 2, host_io=True)
    h = two["host_io"]
    assert h["ranks_reporting"] == 2 and not h["errors"]
    assert h["aggregate_frames_per_s"] > 0 and len(h["per_rank_frames_per_s"]) == 2
    assert h["d2h_GB_per_s_per_gpu"] > 0 and h["h2d_GB_per_s_per_gpu"] > 0


def test_eight_ranks_share_config_4_on_one_gpu():
    """BASELINE config 4 in the shape it names -- 64 independent 1080p SDR pairs over 8 ranks = 8 pairs per rank -- as far as ONE GPU
    allows: eight processes under torch.distributed.run on the single device (gloo reductions).  The job is the same 64 pairs whatever
    the world size, every rank opens device rank % device_count, and rank 0 prints one line."""
    one = run_bench([], 1, "sdr1080_64pairs")
    eight = run_bench([], 8, "sdr1080_64pairs")
    c = eight["config"]
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong"
    assert c["pair_streams_total"] == 64 and c["pair_streams_per_gpu"] == 8 and c["flow_batch"] == 8
    assert c["output_frames_total"] == one["config"]["output_frames_total"]
    assert c["rank_devices"] == [r % c["device_count"] for r in range(8)]
    assert eight["value"] > 0 and "cpu_baseline" not in eight


def test_default_line_carries_the_other_baseline_configs():
    """The driver sees ONE bench line: behind the timed region of the default workload bench.py runs ~1 s legs of BASELINE configs 2, 4
    and 5, of the two sizes round 6 added and of config 1's frame size in child processes and reports them under other_workloads (never
    part of `value`)."""
    d = run_bench(["--no-profile"], 1, "hdr2160_24to120")
    o = dict(d["other_workloads"])
    assert o.pop("failed") == []          # a leg that fails (or does not fit the legs' shared deadline) is named at the top, not hidden in its entry
    assert set(o) == {"sdr1080_24to60", "sdr1080_64pairs", "hdr2160_nb10_blur32", "hdr1080_24to120", "sdr2160_24to60", "sdr360_24to60"}
    for name, w in o.items():
        assert "error" not in w, (name, w)
        assert w["value"] > 0 and 0 < w["frac"] < 1 and w["frac_algorithmic"] > 0 and w["timed_region_s"] > 0.3, (name, w)
    assert o["sdr1080_64pairs"]["pair_streams"] == 64 and o["sdr1080_64pairs"]["flow_batch"] == 16
    # ... and the other content classes at both sizes, each with the kernels' own counters
    content = dict(d["content"])
    assert content.pop("failed") == [] and content.pop("note")
    assert set(content) == {"hdr2160_24to120", "sdr1080_24to60"}
    for wl, legs in content.items():
        assert set(legs) == {"bench", "bench_wrap6", "static", "pan64", "chaotic", "cut"}, (wl, sorted(legs))
        for scene, leg in legs.items():
            assert leg["value"] > 0 and leg["us_per_flow_calc_in_pipeline"] > 0, (wl, scene, leg)
            cc = leg["counters"]
            if cc["sad_tables"]:
                assert set(cc["reuse_share"]) == {"32", "16", "8", "4", "2"} and cc["reuse_share"]["32"] == {"X": 0.0, "Y": 0.0}
            else:     # hardly any window keeps its offsets: the chains of this content run without the tables (hf_calc.hip choose_tab_mode)
                assert cc["reuse_share"] == {} and cc["still_share_of_32_windows"] < 0.45
        assert all(v == {"X": 1.0, "Y": 1.0} for k, v in legs["static"]["counters"]["reuse_share"].items() if k != "32")
        assert [legs[x]["counters"]["sad_tables"] for x in ("bench", "static", "pan64", "chaotic", "cut")] == [1, 1, 1, 0, 0]
        assert legs["bench"]["counters"]["reuse_share"]["16"]["X"] > 0.8
    assert content["hdr2160_24to120"]["static"]["counters"]["warp_staged_share"] > 0.95 > content["hdr2160_24to120"]["cut"]["counters"]["warp_staged_share"]
    assert d["config"]["scene"] == "bench" and d["config"]["pool_order"] == "pingpong"
    r = d["roofline"]
    assert r["kernel"].startswith("warp_wg_kernel<unsigned short, 2,")
    # the line describes itself: compulsory bytes, both readings of FETCH_SIZE on the chain's gathers, and the device
    assert 0 < r["frac_compulsory"] < r["frac_narrow_gathers_x1"] <= r["frac"] < 1 and r["moved_over_compulsory"] > 1
    assert r["compulsory_bytes_per_pair_and_period"] == int(2 * 24883200 + 5.00625 * (24883200 + 4 * 129600) + 6 * 129600 * 2 + 4 * 129600) or abs(
        r["compulsory_bytes_per_pair_and_period"] / (7 * 24883200) - 1) < 0.03
    dev = d["device"]
    assert dev["name"] and dev["box_id"] and dev["compute_units"] == 256
    assert 300.0 < dev["shader_clock_mhz_under_load"] < 2600.0 and dev["clock_probe_call_ms"] < 5000.0
    assert 2000.0 < dev["hbm_streams_idle_device"]["copy_GBps_read_plus_write"] < 8000.0 and dev["hbm_streams_idle_device"]["fill_GBps"] > 2000.0
    for when in ("at_start_of_timed_region", "at_end_of_timed_region"):
        assert dev[when] is None or set(dev[when]) <= {"sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w", "power_cap_w", "temp_c"}
