"""Sharding of independent frame-pair work across the GPUs of a node.

calculateOpticalFlow has no temporal state (the offset array is zeroed on every call, reference
opticalFlowCalcSDR.cpp:68-69), so the source periods of a timeline -- or whole independent clips --
split across ranks with ZERO exchange: no collective is on the data path.  The only cross-period state is
host-side: the blending phase (precomputed by the planner) and the m_totalFrameDelta history behind the filter's
scene-change decision (HopperRender.cpp:1126-1176: the average of up to 10 earlier periods + the next one).  The deltas
are GPU results of the PRECEDING rank's periods, so a chunk starts `delta_history` periods early and replays only their
flow calculations (no warps, no output) -- cheaper than exchanging the history, and it keeps the ranks independent.

Two partitions are offered:
  * shard_clips:     independent clips / pair streams, round-robin (BASELINE config 4: 64 pairs over 8 GPUs)
  * shard_timeline:  one long clip cut into contiguous chunks; chunk g needs the frames before its first period
                     as overlap (ring N-2, N-1, N + previous flow = 3, + 12 periods of delta history when the
                     scene-change detector is on); run_chunk() executes one chunk with the native protocol state
Results are gathered in index order by the caller (rank-local lists; torch.distributed only for the
barrier/timing in bench.py).
"""
from dataclasses import dataclass
from typing import List

from .protocol import SOURCE_24, TARGET_60, BlendSchedule


def shard_clips(n_clips: int, world: int, rank: int) -> List[int]:
    """Indices of the clips / pair streams rank `rank` owns (round-robin)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return list(range(rank, n_clips, world))


@dataclass
class TimelineChunk:
    first_period: int        # index of the first source period (= source frame) this rank interpolates
    n_periods: int
    first_frame: int         # first source frame to upload (includes the overlap frames)
    n_frames: int
    scalars: List[List[float]]  # blending scalars per owned period (the filter's exact schedule)
    first_output: int        # global index of this chunk's first output frame
    blend_at_start: float = 0.0  # m_dBlendingScalar when the first owned period begins


DELTA_HISTORY = 12   # periods the scene-change decision looks back: average of <= 10 + current + next (HopperRender.cpp:1131-1144)


def shard_timeline(n_source_frames: int, world: int, rank: int, source_frame_time=SOURCE_24,
                   target_frame_time=TARGET_60, overlap=3, delta_history=DELTA_HISTORY) -> TimelineChunk:
    """Contiguous chunk of the source timeline for `rank`.  Periods before the third frame of the clip only
    copy frames (HopperRender.cpp:955,1179), which rank 0 keeps.  `overlap` = frames needed before a period
    so that the ring holds N-2, N-1, N AND the previous flow exists (3); `delta_history` = further periods whose
    flow is replayed so that the delta history equals the sequential run's (0: scene-change detection off)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    sched = BlendSchedule(source_frame_time, target_frame_time)
    plan, blends = [], []
    for _ in range(n_source_frames):
        blends.append(sched.blend)
        n = sched.begin_source_frame()
        plan.append([sched.next_scalar() for _ in range(n)])
    base, rem = divmod(n_source_frames, world)
    start = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    first_frame = max(0, start - overlap - delta_history)
    first_output = sum(len(p) for p in plan[:start])
    return TimelineChunk(start, count, first_frame, start + count - first_frame, plan[start:start + count], first_output,
                         blends[start] if start < n_source_frames else 0.0)


def run_chunk(calc, chunk: TimelineChunk, frames, frame_output=2, scene_change_threshold=None, source_frame_time=SOURCE_24,
              target_frame_time=TARGET_60):
    """One rank's share of a clip, output for output what the sequential filter produces for those periods:
    `frames[k]` = source frame k of the WHOLE clip (only chunk.first_frame .. are touched).  The frames before the first
    owned period warm up the ring, the previous flow and the delta history; the owned periods run the filter's
    protocol (native hf_filter state) with the planner's blending scalars.  Returns (output frames, kinds)."""
    from .protocol import DEFAULT_SCENE_CHANGE_THRESHOLD, NativeFilter
    thr = DEFAULT_SCENE_CHANGE_THRESHOLD if scene_change_threshold is None else scene_change_threshold
    host = NativeFilter(source_frame_time, target_frame_time, frame_output, thr)
    outs, kinds = [], []
    for k in range(chunk.first_frame, chunk.first_frame + chunk.n_frames):
        calc.updateFrame(frames[k])
        count = k + 1                                   # the sequential run's m_frameCount at this frame
        if calc.m_frameCount >= 3:
            calc.calculateOpticalFlow()
            host.push(count, calc.m_totalFrameDelta)
        if k < chunk.first_period:
            continue                                    # warm-up: ring, previous flow, delta history -- no output
        for t in chunk.scalars[k - chunk.first_period]:
            cut = host.detect(count)
            if count >= 3 and not cut:
                calc.warpFrames(t, frame_output); kinds.append("warp")
            else:
                calc.copyFrame(); kinds.append("copy")
            outs.append(calc.downloadFrame().copy())
    host.close()
    return outs, kinds
