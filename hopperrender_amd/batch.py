"""Sharding of independent frame-pair work across the GPUs of a node.

calculateOpticalFlow has no temporal state (the offset array is zeroed on every call, reference
opticalFlowCalcSDR.cpp:68-69), so the source periods of a timeline -- or whole independent clips --
split across ranks with ZERO exchange: no collective is on the data path.  The only cross-period state is
host-side (blending phase, delta history), which the planner precomputes.

Two partitions are offered:
  * shard_clips:     independent clips / pair streams, round-robin (BASELINE config 4: 64 pairs over 8 GPUs)
  * shard_timeline:  one long clip cut into contiguous chunks; chunk g needs the two frames before its
                     first period as overlap (flow N-1->N and the warp's frames N-2, N-1)
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


def shard_timeline(n_source_frames: int, world: int, rank: int, source_frame_time=SOURCE_24,
                   target_frame_time=TARGET_60, overlap=3) -> TimelineChunk:
    """Contiguous chunk of the source timeline for `rank`.  Periods before the third frame of the clip only
    copy frames (HopperRender.cpp:955,1179), which rank 0 keeps.  `overlap` = frames needed before a period
    so that the ring holds N-2, N-1, N AND the previous flow exists (3)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    plan = BlendSchedule(source_frame_time, target_frame_time).plan(n_source_frames)
    base, rem = divmod(n_source_frames, world)
    start = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    first_frame = max(0, start - overlap)
    first_output = sum(len(p) for p in plan[:start])
    return TimelineChunk(start, count, first_frame, start + count - first_frame, plan[start:start + count], first_output)
