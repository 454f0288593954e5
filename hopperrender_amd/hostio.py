"""Python face of the native host-I/O driver of one rank (csrc/hf_hostio.cpp, `hf_hostio_*` / `hf_shard_timeline` in
include/hopperflow.h): frames enter and leave through HOST memory ("one host thread + 3
streams + pinned ring per device, results gathered in index order").

The reference moves every frame with blocking transfers on the streaming thread (updateFrame, opticalFlowCalcSDR.cpp:19-29;
downloadFrame, :31-42).  Here one rank = one process = one GPU runs its share of a clip (batch.TimelineChunk) with

    pinned input ring  --hf_update_frame_async (H2D side stream)-->  3-frame ring + phase plane
    flow chain on the context's stream, warps on its second stream (HF_FLAG_DUAL_STREAM)
    output frames      --hf_download_frame_async (D2H side stream)--> pinned output ring --> sink(index, frame), in index order

and the filter's protocol state (native hf_filter: blending schedule is the planner's, scene-change decision from the
m_totalFrameDelta stream).  The only host wait inside a period is hf_wait_flow -- the decision warp vs copy needs THIS period's
frame delta (HopperRender.cpp:959-972,1126-1176) -- while uploads and readbacks keep running on their streams.  Outputs are
handed to `sink` strictly in order; a slot of the output ring is drained (hf_wait_download) right before it is reused.

Same output frames as batch.run_chunk (blocking) and, chunk by chunk, as the sequential filter (tests/test_hostio_gpu.py).
"""
import ctypes as C

import numpy as np

from . import capi
from .calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
from .protocol import DEFAULT_SCENE_CHANGE_THRESHOLD, SOURCE_24, TARGET_60


def shard_timeline_native(n_source_frames, world, rank, source_frame_time=SOURCE_24, target_frame_time=TARGET_60, overlap=3,
                          delta_history=12):
    """hf_shard_timeline: (HfTimelineChunk, outputs per owned period, blending scalars) -- the planner a C host uses;
    batch.shard_timeline is its Python twin (tests/test_protocol_and_sharding.py compares them)."""
    lib = capi.load()
    ch = capi.HfTimelineChunk()
    args = (int(n_source_frames), int(world), int(rank), int(source_frame_time), int(target_frame_time), int(overlap), int(delta_history))
    rc = lib.hf_shard_timeline(*args, C.byref(ch), None, None, 0)
    if rc != 0:
        raise capi.HopperFlowError(rc, (lib.hf_hostio_last_error(None) or b"").decode())
    n_out = (C.c_int32 * max(1, ch.n_periods))()
    t = (C.c_float * max(1, ch.n_outputs))()
    rc = lib.hf_shard_timeline(*args, C.byref(ch), n_out, t, ch.n_outputs)
    if rc != 0:
        raise capi.HopperFlowError(rc, (lib.hf_hostio_last_error(None) or b"").decode())
    return ch, n_out, t


class HostIoRunner:
    """One asynchronous context + the native driver (hf_hostio_*, csrc/hf_hostio.cpp) with its pinned rings.
    `fill(k, array)` writes source frame k of the clip into `array` (a view of a page-locked buffer: a file reader reads
    straight into it); `sink(i, array, kind)` receives the rank's i-th output frame ('warp' | 'copy'; valid only during the call)."""

    def __init__(self, hdr, height, width, *, device_index=0, delta_scalar=8, neighbor_scalar=6, black=0.0, white=255.0,
                 search_radius=16, blur_radius=0, in_ring=3, out_ring=12):
        cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
        self.calc = cls(height, width, 0, 0, delta_scalar, neighbor_scalar, black, white, 270, device_index=device_index,
                        search_radius=search_radius, blur_radius=blur_radius, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_DUAL_STREAM)
        self._lib = capi.load()
        self.in_ring, self.out_ring = int(in_ring), int(out_ring)
        dt = np.dtype(self.calc.dtype)
        self.n_in = self.calc.input_frame_bytes // dt.itemsize
        self.n_out = self.calc.output_frame_bytes // dt.itemsize
        self.bytes_in = self.bytes_out = 0

    def run(self, chunk, fill, sink, frame_output=2, scene_change_threshold=None, source_frame_time=SOURCE_24,
            target_frame_time=TARGET_60):
        """The rank's chunk of the clip (a batch.TimelineChunk); returns the kinds ('warp' | 'copy') of its output frames."""
        lib, c = self._lib, self.calc
        thr = DEFAULT_SCENE_CHANGE_THRESHOLD if scene_change_threshold is None else scene_change_threshold
        cfg = capi.HfHostioConfig(C.sizeof(capi.HfHostioConfig), self.in_ring, self.out_ring, int(frame_output), int(thr), 0,
                                  int(source_frame_time), int(target_frame_time))
        io = C.c_void_p()
        rc = lib.hf_hostio_create(c._ctx, C.byref(cfg), C.byref(io))
        if rc != 0:
            raise capi.HopperFlowError(rc, (lib.hf_hostio_last_error(None) or b"").decode())
        n_outputs = sum(len(ts) for ts in chunk.scalars)
        ch = capi.HfTimelineChunk(chunk.first_period, chunk.n_periods, chunk.first_frame, chunk.n_frames, chunk.first_output, n_outputs,
                                  chunk.blend_at_start)
        counts = (C.c_int32 * max(1, chunk.n_periods))(*[len(ts) for ts in chunk.scalars])
        t = (C.c_float * max(1, n_outputs))(*[float(x) for ts in chunk.scalars for x in ts])
        kinds = (C.c_int32 * max(1, n_outputs))()
        dt = self.calc.dtype
        failure = []

        def view(ptr, n):
            return np.ctypeslib.as_array((C.c_ubyte * (n * np.dtype(dt).itemsize)).from_address(ptr)).view(dt)

        def c_fill(_user, k, ptr):
            try:
                fill(int(k), view(ptr, self.n_in))
                return 0
            except BaseException as e:        # an exception must not unwind through the C frames
                failure.append(e)
                return 1

        def c_sink(_user, i, ptr, kind):
            try:
                sink(int(i), view(ptr, self.n_out), "warp" if kind else "copy")
                return 0
            except BaseException as e:
                failure.append(e)
                return 1

        cb_fill, cb_sink = capi.HF_HOSTIO_FILL_FN(c_fill), capi.HF_HOSTIO_SINK_FN(c_sink)
        try:
            rc = lib.hf_hostio_run(io, C.byref(ch), counts, t, cb_fill, cb_sink, None, kinds)
            if failure:
                raise failure[0]
            if rc != 0:
                raise capi.HopperFlowError(rc, (lib.hf_hostio_last_error(io) or b"").decode())
            bi, bo = C.c_uint64(), C.c_uint64()
            lib.hf_hostio_get_traffic(io, C.byref(bi), C.byref(bo))
            self.bytes_in += bi.value
            self.bytes_out += bo.value
        finally:
            lib.hf_hostio_destroy(io)
        return ["warp" if kinds[i] else "copy" for i in range(n_outputs)]

    def close(self):
        self.calc.close()
