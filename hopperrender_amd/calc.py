"""Python mirror of the reference's OpticalFlowCalc interface (reference HopperRender/
opticalFlowCalc.h:24-138, opticalFlowCalcSDR.h:10-56, opticalFlowCalcHDR.h:10-56) on the C ABI.

Same constructor arguments, same method names (updateFrame / downloadFrame / calculateOpticalFlow /
warpFrames / copyFrame) and the same public field names (m_*), so tests read like calls made by the
reference's filter (HopperRender.cpp:907-1189).  `init` and `blendFrames` are the aliases
BASELINE.json's north_star uses for the constructor body and warpFrames(t, 2).
Errors surface as HopperFlowError (reference: std::runtime_error).
"""
import ctypes as C

import numpy as np

from . import capi

BlendedFrame = 2  # HopperRender.h:10-18
WarpedFrame12, WarpedFrame21, HSVFlow, GreyFlow, SideBySide1, SideBySide2 = 0, 1, 3, 4, 5, 6


def _ptr(a):
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("frame buffers must be C-contiguous")
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a)


class OpticalFlowCalc:
    """Abstract base in the reference; here the shared implementation (is_hdr set by subclasses)."""

    is_hdr = False

    def __init__(self, frameHeight, frameWidth, inputStride=0, outputStride=0, deltaScalar=8, neighborScalar=6,
                 blackLevel=0.0, whiteLevel=255.0, maxCalcRes=270, *, device_index=0, iterations=0, blur_radius=0,
                 search_radius=0, flags=0):
        self._lib = capi.load()
        self._ctx = C.c_void_p()
        self.init(frameHeight, frameWidth, inputStride, outputStride, deltaScalar, neighborScalar, blackLevel,
                  whiteLevel, maxCalcRes, device_index=device_index, iterations=iterations, blur_radius=blur_radius,
                  search_radius=search_radius, flags=flags)

    def init(self, frameHeight, frameWidth, inputStride, outputStride, deltaScalar, neighborScalar, blackLevel,
             whiteLevel, maxCalcRes, *, device_index=0, iterations=0, blur_radius=0, search_radius=0, flags=0):
        """Constructor body (opticalFlowCalcSDR.cpp:206-325)."""
        if self._ctx:
            self._lib.hf_destroy(self._ctx)
            self._ctx = C.c_void_p()
        cfg = capi.HfConfig(C.sizeof(capi.HfConfig), int(self.is_hdr), frameHeight, frameWidth, inputStride, outputStride,
                            deltaScalar, neighborScalar, blackLevel, whiteLevel, maxCalcRes, device_index, iterations,
                            blur_radius, search_radius, flags)
        capi.check(self._lib.hf_create(C.byref(cfg), C.byref(self._ctx)))
        self.device_index = self._lib.hf_get_device(self._ctx)   # device_index = -1: the first suitable device
        capi._devices_used.add(self.device_index)
        st = self._stats()
        self.m_frameWidth, self.m_frameHeight = st.frame_width, st.frame_height
        self.m_inputStride, self.m_outputStride = st.input_stride, st.output_stride
        self.m_opticalFlowResScalar = st.res_scalar
        self.m_opticalFlowFrameWidth, self.m_opticalFlowFrameHeight = st.low_width, st.low_height
        self.input_frame_bytes, self.output_frame_bytes = st.input_frame_bytes, st.output_frame_bytes
        self.phase_plane_bytes = st.phase_plane_bytes
        self.dtype = np.uint16 if self.is_hdr else np.uint8

    # ---- public fields of the reference object, backed by the context ----
    def _params(self):
        p = capi.HfParams()
        capi.check(self._lib.hf_get_params(self._ctx, C.byref(p)), self._ctx)
        return p

    def _set(self, **kw):
        p = self._params()
        for k, v in kw.items():
            setattr(p, k, v)
        capi.check(self._lib.hf_set_params(self._ctx, C.byref(p)), self._ctx)

    def _stats(self):
        s = capi.HfStats()
        capi.check(self._lib.hf_get_stats(self._ctx, C.byref(s)), self._ctx)
        return s

    m_deltaScalar = property(lambda s: s._params().delta_scalar, lambda s, v: s._set(delta_scalar=int(v)))
    m_neighborBiasScalar = property(lambda s: s._params().neighbor_scalar, lambda s, v: s._set(neighbor_scalar=int(v)))
    m_outputBlackLevel = property(lambda s: s._params().black_level, lambda s, v: s._set(black_level=float(v)))
    m_outputWhiteLevel = property(lambda s: s._params().white_level, lambda s, v: s._set(white_level=float(v)))
    m_opticalFlowSearchRadius = property(lambda s: s._params().search_radius, lambda s, v: s._set(search_radius=int(v)))
    m_frameCount = property(lambda s: s._params().frame_count, lambda s, v: s._set(frame_count=int(v)))
    m_totalFrameDelta = property(lambda s: s._stats().total_frame_delta)
    m_ofcCalcTime = property(lambda s: s._stats().ofc_calc_time)
    m_ofcAvgCalcTime = property(lambda s: s._stats().ofc_avg_calc_time)
    m_ofcPeakCalcTime = property(lambda s: s._stats().ofc_peak_calc_time)
    m_warpCalcTime = property(lambda s: s._stats().warp_calc_time)

    # ---- the five virtuals ----
    def updateFrame(self, inputPlanes):
        """opticalFlowCalcSDR.cpp:19-29.  inputPlanes: host ndarray (NV12 uint8 / P010 uint16)."""
        a = np.ascontiguousarray(inputPlanes)
        if a.nbytes < self.input_frame_bytes:
            raise ValueError(f"input frame has {a.nbytes} bytes, need {self.input_frame_bytes}")
        capi.check(self._lib.hf_update_frame(self._ctx, _ptr(a)), self._ctx)

    def downloadFrame(self, outputPlanes=None):
        """opticalFlowCalcSDR.cpp:31-42.  Returns the flat output frame (allocated if not given)."""
        if outputPlanes is None:
            outputPlanes = np.empty(self.output_frame_bytes // np.dtype(self.dtype).itemsize, dtype=self.dtype)
        if outputPlanes.nbytes < self.output_frame_bytes:
            raise ValueError("output buffer too small")
        capi.check(self._lib.hf_download_frame(self._ctx, _ptr(outputPlanes)), self._ctx)
        return outputPlanes

    def calculateOpticalFlow(self):
        """opticalFlowCalcSDR.cpp:44-139."""
        capi.check(self._lib.hf_calculate_optical_flow(self._ctx), self._ctx)

    def warpFrames(self, blendingScalar, frameOutputMode):
        """opticalFlowCalcSDR.cpp:141-168."""
        capi.check(self._lib.hf_warp_frames(self._ctx, float(blendingScalar), int(frameOutputMode)), self._ctx)

    def blendFrames(self, blendingScalar):
        self.warpFrames(blendingScalar, BlendedFrame)

    def copyFrame(self):
        """opticalFlowCalcSDR.cpp:170-183."""
        capi.check(self._lib.hf_copy_frame(self._ctx), self._ctx)

    # ---- device-resident / async extensions ----
    def updateFrameDevice(self, dev_ptr):
        capi.check(self._lib.hf_update_frame_device(self._ctx, C.c_void_p(dev_ptr)), self._ctx)

    def updateFrameDeviceRef(self, dev_ptr):
        """Zero-copy: the ring references the caller's device frame (valid until 3 more updates)."""
        capi.check(self._lib.hf_update_frame_device_ref(self._ctx, C.c_void_p(dev_ptr)), self._ctx)

    def interpolatePeriod(self, dev_frame_ptr, scalars, out_ptrs, mode=BlendedFrame):
        """updateFrame(ref) + calculateOpticalFlow + one warpFrames per scalar, in ONE native call."""
        n = len(scalars)
        ts = (C.c_float * n)(*[float(x) for x in scalars])
        outs = (C.c_void_p * n)(*[int(p) for p in out_ptrs[:n]])
        capi.check(self._lib.hf_interpolate_period(self._ctx, C.c_void_p(dev_frame_ptr or 0), n, ts, outs, int(mode)), self._ctx)

    def interpolateOnly(self, scalars, out_ptrs, mode=BlendedFrame):
        """The warps of one period (one fused launch when eligible), without updateFrame / calculateOpticalFlow."""
        n = len(scalars)
        ts = (C.c_float * n)(*[float(x) for x in scalars])
        outs = (C.c_void_p * n)(*[int(p) for p in out_ptrs[:n]])
        capi.check(self._lib.hf_interpolate_period_ex(self._ctx, None, n, ts, outs, int(mode), 0), self._ctx)

    def updateFrameAsync(self, pinned):
        """H2D on a side stream; `pinned` = PinnedArray (or its .array) that stays valid until sync()."""
        a = pinned.array if hasattr(pinned, "array") else pinned
        capi.check(self._lib.hf_update_frame_async(self._ctx, _ptr(a)), self._ctx)

    def downloadFrameAsync(self, pinned):
        """D2H of the frame just produced on a side stream; read `pinned` only after sync()."""
        a = pinned.array if hasattr(pinned, "array") else pinned
        capi.check(self._lib.hf_download_frame_async(self._ctx, _ptr(a)), self._ctx)

    def waitFlow(self):
        """Asynchronous contexts: block until the last calculateOpticalFlow has finished (m_totalFrameDelta valid); side streams keep running."""
        capi.check(self._lib.hf_wait_flow(self._ctx), self._ctx)

    def downloadsIssued(self):
        return int(self._lib.hf_downloads_issued(self._ctx))

    def waitDownload(self, index):
        """Block until the index-th downloadFrameAsync (issue order, from 0) has landed in its host buffer."""
        capi.check(self._lib.hf_wait_download(self._ctx, int(index)), self._ctx)

    def downloadFrameDevice(self, dev_ptr):
        capi.check(self._lib.hf_download_frame_device(self._ctx, C.c_void_p(dev_ptr)), self._ctx)

    def setOutputBuffer(self, dev_ptr):
        capi.check(self._lib.hf_set_output_buffer(self._ctx, C.c_void_p(dev_ptr or 0)), self._ctx)

    def sync(self):
        capi.check(self._lib.hf_sync(self._ctx), self._ctx)

    def timerBegin(self):
        capi.check(self._lib.hf_timer_begin(self._ctx), self._ctx)

    def timerEnd(self):
        ms = C.c_float()
        capi.check(self._lib.hf_timer_end(self._ctx, C.byref(ms)), self._ctx)
        return ms.value

    def profile(self):
        """Device-time totals since resetProfile() (needs flags=HF_FLAG_PROFILE)."""
        pr = capi.HfProfile()
        capi.check(self._lib.hf_get_profile(self._ctx, C.byref(pr)), self._ctx)
        return {k: getattr(pr, k) for k, _ in pr._fields_}

    def setProfileInterval(self, warp_every, flow_every):
        capi.check(self._lib.hf_set_profile_interval(self._ctx, int(warp_every), int(flow_every)), self._ctx)

    def resetProfile(self):
        capi.check(self._lib.hf_reset_profile(self._ctx), self._ctx)

    # ---- parity taps ----
    def readOffsets(self):
        a = np.empty((2, self.m_opticalFlowFrameHeight, self.m_opticalFlowFrameWidth), dtype=np.int16)
        capi.check(self._lib.hf_read_offsets(self._ctx, _ptr(a)), self._ctx)
        return a

    def readBlurredFlow(self, idx):
        a = np.empty((2, self.m_opticalFlowFrameHeight, self.m_opticalFlowFrameWidth), dtype=np.int16)
        capi.check(self._lib.hf_read_blurred_flow(self._ctx, idx, _ptr(a)), self._ctx)
        return a

    def readPhasePlane(self, ring_slot):
        """(plane as uint32 words, complete?) -- hf_read_phase_plane"""
        import ctypes
        n = int(self.phase_plane_bytes)
        a = np.empty(n // 4, dtype=np.uint32)
        done = ctypes.c_int(0)
        capi.check(self._lib.hf_read_phase_plane(self._ctx, ring_slot, _ptr(a), ctypes.byref(done)), self._ctx)
        return a, bool(done.value)

    def writeBlurredFlow(self, idx, flow):
        a = np.ascontiguousarray(flow, dtype=np.int16)
        assert a.shape == (2, self.m_opticalFlowFrameHeight, self.m_opticalFlowFrameWidth)
        capi.check(self._lib.hf_write_blurred_flow(self._ctx, idx, _ptr(a)), self._ctx)

    def deviceRcp(self, values):
        x = np.ascontiguousarray(values, dtype=np.float32)
        y = np.empty_like(x)
        capi.check(self._lib.hf_device_rcp(self._ctx, _ptr(x), _ptr(y), len(x)), self._ctx)
        return y

    def countersEnable(self, on=True):
        """hopperflow_diag.h hf_debug_counters_enable: device-side counters of the kernels' per-window / per-workgroup decisions."""
        capi.check(self._lib.hf_debug_counters_enable(self._ctx, 1 if on else 0), self._ctx)

    def counters(self, reset=False):
        """{'warp_workgroups': {staged, interior_global, generic}, 'levels': {window: {'X': (windows, reused), 'Y': ...}}}"""
        cc = capi.HfDebugCounters()
        capi.check(self._lib.hf_debug_counters_read(self._ctx, C.byref(cc), 1 if reset else 0), self._ctx)
        lv = {}
        for k in range(16):
            if cc.level_window_size[k] and (cc.level_windows[k][0] or cc.level_windows[k][1]):
                lv[int(cc.level_window_size[k])] = {"X": (int(cc.level_windows[k][0]), int(cc.level_reused[k][0])),
                                                     "Y": (int(cc.level_windows[k][1]), int(cc.level_reused[k][1]))}
        return {"warp_workgroups": dict(zip(("staged", "interior_global", "generic"), [int(x) for x in cc.warp_workgroups])), "levels": lv}

    def stats(self):
        s = self._stats()
        return {k: getattr(s, k) for k, _ in s._fields_}

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.hf_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OpticalFlowCalcSDR(OpticalFlowCalc):
    """NV12, 8-bit (opticalFlowCalcSDR.h:10-56)."""
    is_hdr = False


class OpticalFlowCalcHDR(OpticalFlowCalc):
    """P010, 16-bit (opticalFlowCalcHDR.h:10-56)."""
    is_hdr = True


class DeviceBuffer:
    """A plain hipMalloc'ed buffer through the C ABI helpers (bench / batch staging)."""

    def __init__(self, nbytes, device_index=0):
        self._lib = capi.load()
        self.device_index, self.nbytes = device_index, nbytes
        p = C.c_void_p()
        capi.check(self._lib.hf_device_malloc(device_index, nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        capi.check(self._lib.hf_memcpy_h2d(self.device_index, C.c_void_p(self.ptr), _ptr(a), a.nbytes))

    def download(self, dtype, count=None):
        dt = np.dtype(dtype)
        n = count if count is not None else self.nbytes // dt.itemsize
        a = np.empty(n, dtype=dt)
        capi.check(self._lib.hf_memcpy_d2h(self.device_index, _ptr(a), C.c_void_p(self.ptr), a.nbytes))
        return a

    def free(self):
        if self.ptr:
            self._lib.hf_device_free(self.device_index, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """A numpy view of page-locked host memory (hf_host_malloc_pinned) for frame I/O at full PCIe rate."""

    def __init__(self, count, dtype):
        self._lib = capi.load()
        dt = np.dtype(dtype)
        p = C.c_void_p()
        capi.check(self._lib.hf_host_malloc_pinned(count * dt.itemsize, C.byref(p)))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_ubyte * (count * dt.itemsize)).from_address(self.ptr)).view(dt)

    def free(self):
        if self.ptr:
            self.array = None
            self._lib.hf_host_free_pinned(C.c_void_p(self.ptr))
            self.ptr = None


class FlowBatch:
    """hf_batch: calculateOpticalFlow() of up to 32 contexts (independent frame pairs, same geometry and parameters)
    as one set of launches.  While the batch exists its members issue on one stream; close() it before them."""

    def __init__(self, members):
        self._lib = capi.load()
        self.members = list(members)
        arr = (C.c_void_p * len(self.members))(*[m._ctx for m in self.members])
        out = C.c_void_p()
        rc = self._lib.hf_batch_create(arr, len(self.members), C.byref(out))
        if rc != 0:
            raise capi.HopperFlowError(rc, (self._lib.hf_batch_last_error(None) or b"").decode())
        self._b = out

    def defersPlanes(self):
        """hf_batch_defers_planes: runPeriod samples only the grid of a new frame, the next period's warp launch builds its plane"""
        return bool(self._lib.hf_batch_defers_planes(self._b))

    def calculateOpticalFlow(self):
        self._check(self._lib.hf_batch_calculate_optical_flow(self._b))

    def _check(self, rc):
        if rc != 0:
            raise capi.HopperFlowError(rc, (self._lib.hf_batch_last_error(self._b) or b"").decode())

    def updateFramesDeviceRef(self, dev_ptrs):
        """updateFrameDeviceRef of every member; the phase planes of all new frames are built by one launch."""
        arr = (C.c_void_p * len(self.members))(*[int(p) for p in dev_ptrs])
        self._check(self._lib.hf_batch_update_frames_device_ref(self._b, arr))

    def interpolatePeriod(self, scalars, out_ptrs, mode=BlendedFrame):
        """The warps of one source period of every member in ONE launch.  scalars[i] / out_ptrs[i]: member i's lists."""
        n, K = len(self.members), capi.HF_MAX_PERIOD_OUTPUTS
        counts = (C.c_int * n)(*[len(ts) for ts in scalars])
        t = (C.c_float * (n * K))()
        outs = (C.c_void_p * (n * K))()
        for i, ts in enumerate(scalars):
            for k, x in enumerate(ts):
                t[i * K + k] = float(x)
                outs[i * K + k] = int(out_ptrs[i][k])
        self._check(self._lib.hf_batch_interpolate_period(self._b, counts, t, outs, int(mode)))

    def preparePeriod(self, dev_ptrs, scalars, out_ptrs, mode=BlendedFrame, calculate_flow=True):
        """The arguments of one hf_batch_run_period call, marshalled once (a driver's schedule is known ahead of time);
        dev_ptrs / scalars may be None to skip the update / the warps."""
        n, K = len(self.members), capi.HF_MAX_PERIOD_OUTPUTS
        frames = (C.c_void_p * n)(*[int(p) for p in dev_ptrs]) if dev_ptrs is not None else None
        counts = t = outs = None
        if scalars is not None:
            counts = (C.c_int * n)(*[len(ts) for ts in scalars])
            t = (C.c_float * (n * K))()
            outs = (C.c_void_p * (n * K))()
            for i, ts in enumerate(scalars):
                for k, x in enumerate(ts):
                    t[i * K + k] = float(x)
                    outs[i * K + k] = int(out_ptrs[i][k])
        return (frames, 1 if calculate_flow else 0, counts, t, outs, int(mode))

    def runPeriod(self, prepared):
        """updateFramesDeviceRef + calculateOpticalFlow + interpolatePeriod of one source period in ONE native call."""
        rc = self._lib.hf_batch_run_period(self._b, *prepared)
        if rc != 0:
            self._check(rc)

    def sync(self):
        self._check(self._lib.hf_batch_sync(self._b))

    def timelineEnable(self, max_launches, skip_periods=0):
        """hf_batch_timeline_enable: after skip_periods further runPeriod calls every dispatch carries its own start / stop events (0: off)."""
        self._check(self._lib.hf_batch_timeline_enable(self._b, int(max_launches), int(skip_periods)))

    def timelineRead(self):
        """hf_batch_timeline_read: [(kernel, period, start_ms, end_ms)] on the device's clock, relative to the process's reference event."""
        n = C.c_int(0)
        self._check(self._lib.hf_batch_timeline_read(self._b, None, 0, C.byref(n)))
        if n.value == 0:
            return []
        recs = (capi.HfTimelineRecord * n.value)()
        self._check(self._lib.hf_batch_timeline_read(self._b, recs, n.value, C.byref(n)))
        return [(r.kernel.decode(), r.period, r.start_ms, r.start_ms + r.duration_ms) for r in recs[:n.value] if not (r.flags & 1)]

    def __len__(self):
        return self._lib.hf_batch_size(self._b)

    def close(self):
        if self._b:
            self._lib.hf_batch_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

