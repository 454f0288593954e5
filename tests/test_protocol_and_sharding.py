"""CPU: the host logic restated from the reference's filter (blend schedule, scene-change detector,
governor) and the pair-sharding planner, incl. a world_size-2 gloo run."""
import os
import subprocess
import sys

import pytest

from hopperrender_amd import batch
from hopperrender_amd.protocol import (SOURCE_24, TARGET_60, TARGET_120, BlendSchedule, FilterReplay,
                                       SceneChangeDetector)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blend_schedule_24_to_60_matches_survey():
    p = BlendSchedule(SOURCE_24, TARGET_60).plan(4)
    assert [[round(t, 4) for t in x] for x in p] == [[0.0, 0.3996, 0.7992], [0.1988, 0.5984, 0.998], [0.3976, 0.7972], [0.1968, 0.5964, 0.996]]
    p = BlendSchedule(SOURCE_24, TARGET_120).plan(1000)
    assert 5000 <= sum(len(x) for x in p) <= 5010 and all(5 <= len(x) <= 6 for x in p)
    assert BlendSchedule(SOURCE_24, TARGET_60, active=False).plan(3) == [[0.0], [0.0], [0.0]]


def test_scene_change_detector():
    d = SceneChangeDetector(SOURCE_24, threshold=200)
    for i, delta in enumerate([100, 110, 105, 900, 120, 100]):
        d.push(i + 3, delta)
        hit = d.detect(i + 3)
        # the spike (900) is "current" when it sits second-to-last: avg of the previous <= 10 incl. itself
        if i == 4:
            assert hit
        else:
            assert not hit
    assert d.peak_delta > 0
    d.reset()
    assert not d.detect(0)


class FakeCalc:
    """Records the calls FilterReplay makes (no GPU)."""

    def __init__(self):
        self.m_frameCount, self.m_totalFrameDelta, self.m_ofcCalcTime, self.m_warpCalcTime = 0, 100, 0.001, 0.0005
        self.m_opticalFlowSearchRadius = 5
        self.calls = []

    def updateFrame(self, f): self.m_frameCount += 1; self.calls.append("update")
    def calculateOpticalFlow(self): self.calls.append("calc")
    def warpFrames(self, t, m): self.calls.append(("warp", round(t, 4), m))
    def copyFrame(self): self.calls.append("copy")
    def downloadFrame(self): self.calls.append("download"); return b""


def test_filter_replay_call_sequence():
    c = FakeCalc()
    r = FilterReplay(c, SOURCE_24, TARGET_60, auto_adjust=True)
    for k in range(4):
        r.deliver(None)
    # frames 1,2: copy only; from frame 3 on: calc + warps (HopperRender.cpp:955,1179)
    assert c.calls[:5] == ["update", "copy", "download", "copy", "download"]
    assert "calc" not in c.calls[:c.calls.index("update", 1) + 1]
    third = [i for i, x in enumerate(c.calls) if x == "update"][2]
    assert c.calls[third + 1] == "calc" and c.calls[third + 2][0] == "warp"
    assert c.m_opticalFlowSearchRadius > 5          # plenty of headroom: governor raises R (HopperRender.cpp:1454-1458)
    r.new_segment()
    assert c.m_frameCount == 0


def test_sharding_plans():
    assert batch.shard_clips(64, 8, 3) == list(range(3, 64, 8))
    assert sorted(sum((batch.shard_clips(10, 4, r) for r in range(4)), [])) == list(range(10))
    chunks = [batch.shard_timeline(100, 8, r) for r in range(8)]
    assert sum(c.n_periods for c in chunks) == 100
    assert chunks[0].first_period == 0 and all(chunks[i].first_period == chunks[i - 1].first_period + chunks[i - 1].n_periods for i in range(1, 8))
    assert chunks[3].first_frame == chunks[3].first_period - 3 - batch.DELTA_HISTORY      # ring + previous flow + delta history
    assert batch.shard_timeline(100, 8, 3, delta_history=0).first_frame == chunks[3].first_period - 3
    sched = BlendSchedule(SOURCE_24, TARGET_60)
    for k in range(chunks[5].first_period):
        for _ in range(sched.begin_source_frame()):
            sched.next_scalar()
    assert chunks[5].blend_at_start == sched.blend
    total = sum(len(s) for c in chunks for s in c.scalars)
    assert total == sum(len(x) for x in BlendSchedule(SOURCE_24, TARGET_60).plan(100))
    assert chunks[5].first_output == sum(len(s) for c in chunks[:5] for s in c.scalars)


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding(tmp_path):
    """world_size 2 on CPU (gloo): each rank plans its shard, the union covers every pair exactly once and
    the MAX-over-ranks timing reduction of bench.py works."""
    script = tmp_path / "w.py"
    script.write_text(f'''
import os, sys, json
sys.path.insert(0, {ROOT!r})
import torch, torch.distributed as dist
from hopperrender_amd import batch
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
mine = batch.shard_clips(64, w, r)
ch = batch.shard_timeline(50, w, r)
t = torch.tensor([float(r + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
n = torch.tensor([float(len(mine))], dtype=torch.float64)
dist.all_reduce(n, op=dist.ReduceOp.SUM)
gathered = [None] * w
dist.all_gather_object(gathered, (mine, ch.first_period, ch.n_periods))
if r == 0:
    print(json.dumps(dict(tmax=t.item(), total=n.item(), gathered=gathered)))
dist.barrier(); dist.destroy_process_group()
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29631", str(script)], capture_output=True, text=True, env=env, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["tmax"] == 2.0 and j["total"] == 64.0
    pairs = sorted(sum((g[0] for g in j["gathered"]), []))
    assert pairs == list(range(64))
    assert j["gathered"][0][1] == 0 and j["gathered"][1][1] == j["gathered"][0][2]


def test_native_timeline_planner_equals_the_python_planner(native_lib):
    """hf_shard_timeline (csrc/hf_hostio.cpp: what a C host plans its ranks with, the schedule taken from the native hf_filter) ==
    batch.shard_timeline for every rank: chunk bounds, warm-up frames, first output index, outputs per period, blending scalars
    (as the float the warp call receives), blending phase at the chunk start; the chunks tile the output index space."""
    import ctypes as C
    from hopperrender_amd import batch
    from hopperrender_amd.hostio import shard_timeline_native
    from hopperrender_amd.protocol import SOURCE_24, TARGET_60, TARGET_120
    for n, world, target, dh in [(40, 3, TARGET_60, 12), (97, 8, TARGET_120, 12), (5, 4, TARGET_60, 12), (3, 8, TARGET_120, 0), (64, 1, 100000, 12),
                                 (30, 2, 417083, 12), (0, 2, TARGET_60, 12)]:
        nxt = 0
        for rank in range(world):
            want = batch.shard_timeline(n, world, rank, SOURCE_24, target, delta_history=dh)
            ch, n_out, t = shard_timeline_native(n, world, rank, SOURCE_24, target, delta_history=dh)
            assert (ch.first_period, ch.n_periods, ch.first_frame, ch.n_frames, ch.first_output) == \
                   (want.first_period, want.n_periods, want.first_frame, want.n_frames, want.first_output), (n, world, rank)
            assert [n_out[i] for i in range(ch.n_periods)] == [len(ts) for ts in want.scalars]
            flat = [x for ts in want.scalars for x in ts]
            assert ch.n_outputs == len(flat) and [t[i] for i in range(len(flat))] == [C.c_float(x).value for x in flat]
            assert ch.blend_at_start == want.blend_at_start
            assert ch.first_output == nxt
            nxt += ch.n_outputs
    with __import__("pytest").raises(Exception):
        shard_timeline_native(10, 2, 2)
