"""CPU: tools/timeline_report.py on synthetic dispatch records (the analysis behind profiles/r05_pipeline_timeline.json)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from timeline_report import analyze  # noqa: E402


def _stream(offset, n_periods, warp_ms=2.0, chain_ms=(0.1, 0.2), gap_ms=0.01):
    recs, t = [], offset
    for p in range(n_periods):
        recs.append(("warp_period", p, t, t + warp_ms)); t += warp_ms + gap_ms
        for i, d in enumerate(chain_ms):
            recs.append((f"level_{i}", p, t, t + d)); t += d + gap_ms
    return recs


def test_two_queues_in_lockstep_and_one_alone():
    a = _stream(0.0, 10)
    b = _stream(1.0, 10)                     # the same schedule, one millisecond later
    alone = _stream(0.0, 4, warp_ms=1.0, chain_ms=(0.05, 0.1))
    r = analyze([a, b], alone, periods_per_step=64)
    period = 2.0 + 0.1 + 0.2 + 3 * 0.01
    assert abs(r["mean_period_ms_per_queue"] - period) < 1e-6
    assert abs(r["implied_ms_per_step"] - 64 * period) < 1e-3
    for q in r["queues"]:
        assert 0.98 < q["busy_frac"] + q["gap_us"]["sum_ms"] / r["window_ms"] <= 1.001
        assert abs(q["gap_us"]["median"] - 10.0) < 1e-3
    k = r["kernels"]
    assert k["warp_period"]["stretch_vs_alone"] == 2.0 and k["level_1"]["stretch_vs_alone"] == 2.0
    # queue b's 2 ms warp starts 1 ms into a's and ends 0.67 ms into a's next one (period 2.33 ms): 1.67 of its 2 ms beside another warp
    w = k["warp_period"]["time_frac_with_n_other_period_warps_running"]
    assert abs(w[0] + w[1] - 1.0) < 1e-6 and abs(w[1] - 1.67 / 2.0) < 0.01
    c = r["concurrency"]["time_frac_with_n_queues_busy"]
    assert abs(sum(c) - 1.0) < 1e-6 and c[2] > 0.9
    # stand-alone time of a batch period x queues over the wall time of a period: 2 x 1.15 / 2.33
    assert abs(r["serial_over_pipelined"] - 2 * 1150.0 / 1e3 / period) < 0.01


def test_disjoint_windows_are_reported_not_analysed():
    r = analyze([_stream(0.0, 2), _stream(100.0, 2)])
    assert "error" in r
