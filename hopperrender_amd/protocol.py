"""Replay of the reference filter's call protocol around the calculator.

Restated from reference HopperRender/HopperRender.cpp:
  * DeliverToRenderer :938-957,979-991,1126-1197  -- frame-count gating, number of interpolated
    frames, blending-scalar schedule, scene-change decision, warp-vs-copy, download
  * NewSegment :840                                -- m_frameCount reset
  * autoAdjustSettings :1438-1463                  -- search-radius governor
This is host logic only (no pixels); it drives any object with the OpticalFlowCalc interface.
"""
import math
from collections import deque

# config.h
MIN_SEARCH_RADIUS, MAX_SEARCH_RADIUS = 5, 16
UPPER_PERF_BUFFER, LOWER_PERF_BUFFER = 1.4, 1.6
DEFAULT_SCENE_CHANGE_THRESHOLD = 200

# 100-ns units (HopperRender.cpp:162-163 defaults: 23.976 fps source, 60 fps target)
SOURCE_24 = 417083
TARGET_60 = 166667
TARGET_120 = 83333


class BlendSchedule:
    """m_iNumIntFrames / m_dBlendingScalar bookkeeping (HopperRender.cpp:944-948,1192-1197)."""

    def __init__(self, source_frame_time=SOURCE_24, target_frame_time=TARGET_60, active=True):
        self.source, self.target, self.active = source_frame_time, target_frame_time, active
        self.blend = 0.0

    def begin_source_frame(self):
        if self.active:
            return int(max(math.ceil((1.0 - self.blend) / (float(self.target) / float(self.source))), 1.0))
        return 1

    def next_scalar(self):
        """Returns the scalar to warp with, then advances it."""
        t = self.blend
        if self.active:
            self.blend += float(self.target) / float(self.source)
            if self.blend >= 1.0:
                self.blend -= 1.0
        return t

    def plan(self, n_source_frames):
        """List (per source frame) of the blending scalars the filter would use."""
        out = []
        for _ in range(n_source_frames):
            n = self.begin_source_frame()
            out.append([self.next_scalar() for _ in range(n)])
        return out


class SceneChangeDetector:
    """Delta-history based scene-change test (HopperRender.cpp:959-972,1126-1176)."""

    def __init__(self, source_frame_time=SOURCE_24, threshold=DEFAULT_SCENE_CHANGE_THRESHOLD):
        self.source, self.threshold = source_frame_time, threshold
        self.frame_delta_history = deque()      # (frameNumber, totalDelta)
        self.scene_change_history = deque()     # (frameNumber, delta1, delta2)
        self.peak_delta, self.peak_delta2 = 0, 0

    def reset(self):
        self.frame_delta_history.clear()
        self.scene_change_history.clear()
        self.peak_delta = self.peak_delta2 = 0

    def push(self, frame_count, total_delta):
        frames_in_3s = int(3.0 * 10000000.0 / self.source)
        self.frame_delta_history.append((frame_count, total_delta))
        while self.frame_delta_history and (frame_count - self.frame_delta_history[0][0]) > frames_in_3s:
            self.frame_delta_history.popleft()

    def detect(self, frame_count):
        h = self.frame_delta_history
        if len(h) < 3:
            return False
        n = len(h)
        count = min(n - 2, 10)
        s = sum(h[n - 2 - i][1] for i in range(count))
        average = int(s // count)
        nxt, cur = int(h[n - 1][1]), int(h[n - 2][1])
        d1, d2 = cur - average, cur - nxt
        if d1 > 0:
            frames_in_1s = int(1.0 * 10000000.0 / self.source)
            self.scene_change_history.append((frame_count, d1, d2 if d2 > 0 else 0))
            while self.scene_change_history and (frame_count - self.scene_change_history[0][0]) > frames_in_1s:
                self.scene_change_history.popleft()
            self.peak_delta = self.peak_delta2 = 0
            for (_, a, b) in self.scene_change_history:
                if a > self.peak_delta:
                    self.peak_delta, self.peak_delta2 = a, b
        return d1 >= self.threshold and d1 > 0 and d2 >= self.threshold and d2 > 0


class FilterReplay:
    """Drives a calculator the way CHopperRender::DeliverToRenderer does, one source frame at a time."""

    def __init__(self, calc, source_frame_time=SOURCE_24, target_frame_time=TARGET_60, frame_output=2,
                 scene_change_threshold=DEFAULT_SCENE_CHANGE_THRESHOLD, auto_adjust=False):
        self.calc = calc
        self.schedule = BlendSchedule(source_frame_time, target_frame_time, active=True)
        self.detector = SceneChangeDetector(source_frame_time, scene_change_threshold)
        self.frame_output = frame_output
        self.auto_adjust = auto_adjust
        self.total_warp_duration = 0.0
        self.playback_frame_time = source_frame_time
        self.log = []  # (kind, t) per output frame: 'warp' | 'copy'

    def new_segment(self):
        self.calc.m_frameCount = 0        # HopperRender.cpp:840
        self.detector.reset()             # :828-831

    def auto_adjust_settings(self):
        """HopperRender.cpp:1438-1463."""
        frame_time_s = float(self.playback_frame_time) / 10000000.0
        cur = self.calc.m_ofcCalcTime + self.total_warp_duration
        r = self.calc.m_opticalFlowSearchRadius
        if cur * UPPER_PERF_BUFFER > frame_time_s:
            if r > MIN_SEARCH_RADIUS:
                self.calc.m_opticalFlowSearchRadius = r - 1
        elif cur * LOWER_PERF_BUFFER < frame_time_s:
            if r < MAX_SEARCH_RADIUS:
                self.calc.m_opticalFlowSearchRadius = r + 1
        self.total_warp_duration = 0.0

    def deliver(self, input_frame):
        """One source frame in -> list of output frames (host ndarrays), like DeliverToRenderer."""
        c = self.calc
        n_int = self.schedule.begin_source_frame()          # :944-948
        if self.auto_adjust:
            self.auto_adjust_settings()                     # :951
        c.updateFrame(input_frame)                          # :953
        if c.m_frameCount >= 3:                             # :955
            c.calculateOpticalFlow()
            self.detector.push(c.m_frameCount, c.m_totalFrameDelta)
        outs = []
        for _ in range(n_int):
            scene_change = self.detector.detect(c.m_frameCount)
            if c.m_frameCount >= 3 and not scene_change:    # :1179
                t = self.schedule.blend
                c.warpFrames(t, self.frame_output)
                self.log.append(("warp", t))
            else:
                c.copyFrame()
                self.log.append(("copy", self.schedule.blend))
            outs.append(c.downloadFrame())                  # :1186
            self.total_warp_duration += c.m_warpCalcTime    # :1189
            self.schedule.next_scalar()                     # :1192-1197
        return outs
