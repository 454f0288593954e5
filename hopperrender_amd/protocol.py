"""Replay of the reference filter's call protocol around the calculator.

The product implementation of this logic is native (hopperrender_amd/csrc/hf_filter.cpp behind the hf_filter_* C ABI);
`NativeFilter` / `FilterReplay` below are its ctypes mirror.  `BlendSchedule`, `SceneChangeDetector` and
`auto_adjust_radius` are a pure-Python restatement kept as an independent cross-check of the native code (tests) and
as the planner bench.py / batch.py use to precompute blending scalars.

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


def auto_adjust_radius(ofc_calc_time, total_warp_duration, playback_frame_time, radius):
    """autoAdjustSettings (HopperRender.cpp:1438-1463): the new m_opticalFlowSearchRadius."""
    frame_time_s = float(playback_frame_time) / 10000000.0
    cur = ofc_calc_time + total_warp_duration
    if cur * UPPER_PERF_BUFFER > frame_time_s:
        return radius - 1 if radius > MIN_SEARCH_RADIUS else radius
    if cur * LOWER_PERF_BUFFER < frame_time_s:
        return radius + 1 if radius < MAX_SEARCH_RADIUS else radius
    return radius


class NativeFilter:
    """ctypes mirror of hf_filter (include/hopperflow.h "caller protocol")."""

    def __init__(self, source_frame_time=SOURCE_24, target_frame_time=TARGET_60, frame_output=2,
                 scene_change_threshold=DEFAULT_SCENE_CHANGE_THRESHOLD, auto_adjust=False, active=True):
        import ctypes as C
        from . import capi
        self._C, self._capi, self._lib = C, capi, capi.load()
        cfg = capi.HfFilterConfig(C.sizeof(capi.HfFilterConfig), scene_change_threshold, source_frame_time, target_frame_time,
                                  frame_output, int(auto_adjust), int(active), 0)
        self._f = C.c_void_p()
        rc = self._lib.hf_filter_create(C.byref(cfg), C.byref(self._f))
        if rc:
            raise capi.HopperFlowError(rc, "[HopperRender] hf_filter_create failed")

    def new_segment(self, rate=1.0):
        self._lib.hf_filter_new_segment(self._f, float(rate))

    def set_playback_frame_time(self, t):
        self._lib.hf_filter_set_playback_frame_time(self._f, int(t))

    def begin_source_frame(self):
        return self._lib.hf_filter_begin_source_frame(self._f)

    @property
    def blend(self):
        return self._lib.hf_filter_blending_scalar(self._f)

    def next_scalar(self):
        t = self.blend
        self._lib.hf_filter_advance_blending_scalar(self._f)
        return t

    def add_warp_duration(self, seconds):
        self._lib.hf_filter_add_warp_duration(self._f, float(seconds))

    def auto_adjust(self, ofc_calc_time, radius):
        r = self._C.c_int32(int(radius))
        self._lib.hf_filter_auto_adjust(self._f, float(ofc_calc_time), self._C.byref(r))
        return r.value

    def push(self, frame_count, total_delta):
        self._lib.hf_filter_push_frame_delta(self._f, int(frame_count), int(total_delta))

    def detect(self, frame_count):
        return bool(self._lib.hf_filter_detect_scene_change(self._f, int(frame_count)))

    def state(self):
        st = self._capi.HfFilterState()
        self._lib.hf_filter_get_state(self._f, self._C.byref(st))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def deliver(self, calc, input_frame):
        """hf_filter_deliver: one whole DeliverToRenderer inside the library.  Returns (frames, kinds)."""
        import numpy as np
        C = self._C
        a = np.ascontiguousarray(input_frame)
        # the number of outputs of this period is a pure function of the filter state (m_iNumIntFrames, :944-948; a slowed-down
        # segment can ask for any number): ask first, then hand over exactly that many frame buffers
        K = max(1, self.begin_source_frame())
        outs = [np.empty(calc.output_frame_bytes // np.dtype(calc.dtype).itemsize, dtype=calc.dtype) for _ in range(K)]
        ptrs = (C.c_void_p * K)(*[o.ctypes.data for o in outs])
        n, kinds = C.c_int(), (C.c_int32 * K)()
        self._capi.check(self._lib.hf_filter_deliver(self._f, calc._ctx, a.ctypes.data_as(C.c_void_p), ptrs, K, C.byref(n), kinds), calc._ctx)
        return outs[:n.value], ["warp" if kinds[i] else "copy" for i in range(n.value)]

    def close(self):
        if getattr(self, "_f", None):
            self._lib.hf_filter_destroy(self._f)
            self._f = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FilterReplay:
    """Drives a calculator the way CHopperRender::DeliverToRenderer does, one source frame at a time, with the NATIVE
    protocol state (hf_filter); the calculator calls go through the object's OpticalFlowCalc interface."""

    def __init__(self, calc, source_frame_time=SOURCE_24, target_frame_time=TARGET_60, frame_output=2,
                 scene_change_threshold=DEFAULT_SCENE_CHANGE_THRESHOLD, auto_adjust=False):
        self.calc = calc
        self.host = NativeFilter(source_frame_time, target_frame_time, frame_output, scene_change_threshold, auto_adjust)
        self.frame_output = frame_output
        self.auto_adjust = auto_adjust
        self.log = []  # (kind, t) per output frame: 'warp' | 'copy'

    def new_segment(self, rate=1.0):
        self.calc.m_frameCount = 0        # HopperRender.cpp:840
        self.host.new_segment(rate)       # :836-838

    def deliver(self, input_frame):
        """One source frame in -> list of output frames (host ndarrays), like DeliverToRenderer."""
        c, h = self.calc, self.host
        n_int = h.begin_source_frame()                      # :944-948
        if self.auto_adjust:
            c.m_opticalFlowSearchRadius = h.auto_adjust(c.m_ofcCalcTime, c.m_opticalFlowSearchRadius)   # :951
        c.updateFrame(input_frame)                          # :953
        if c.m_frameCount >= 3:                             # :955
            c.calculateOpticalFlow()
            h.push(c.m_frameCount, c.m_totalFrameDelta)
        outs = []
        for _ in range(n_int):
            scene_change = h.detect(c.m_frameCount)
            t = h.blend
            if c.m_frameCount >= 3 and not scene_change:    # :1179
                c.warpFrames(t, self.frame_output)
                self.log.append(("warp", t))
            else:
                c.copyFrame()
                self.log.append(("copy", t))
            outs.append(c.downloadFrame())                  # :1186
            h.add_warp_duration(c.m_warpCalcTime)           # :1189
            h.next_scalar()                                 # :1192-1197
        return outs
