"""Host-I/O driver of one rank: frames enter and leave through HOST memory ("one host thread + 3
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
import numpy as np

from . import capi
from .calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR, PinnedArray
from .protocol import DEFAULT_SCENE_CHANGE_THRESHOLD, SOURCE_24, TARGET_60, NativeFilter


class HostIoRunner:
    """One context with pinned rings.  `fill(k, array)` writes source frame k of the clip into `array` (a view of pinned
    memory: a file reader reads straight into it); `sink(i, array, kind)` receives the rank's i-th output frame (valid only
    during the call)."""

    def __init__(self, hdr, height, width, *, device_index=0, delta_scalar=8, neighbor_scalar=6, black=0.0, white=255.0,
                 search_radius=16, blur_radius=0, in_ring=3, out_ring=12):
        cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
        self.calc = cls(height, width, 0, 0, delta_scalar, neighbor_scalar, black, white, 270, device_index=device_index,
                        search_radius=search_radius, blur_radius=blur_radius, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_DUAL_STREAM)
        c = self.calc
        dt = c.dtype
        self.n_in = c.input_frame_bytes // np.dtype(dt).itemsize
        self.n_out = c.output_frame_bytes // np.dtype(dt).itemsize
        if in_ring < 3 or out_ring < 2:
            raise ValueError("in_ring >= 3 and out_ring >= 2")
        self.ins = [PinnedArray(self.n_in, dt) for _ in range(in_ring)]
        self.outs = [PinnedArray(self.n_out, dt) for _ in range(out_ring)]
        self.bytes_in = self.bytes_out = 0

    def run(self, chunk, fill, sink, frame_output=2, scene_change_threshold=None, source_frame_time=SOURCE_24,
            target_frame_time=TARGET_60):
        """The rank's chunk of the clip; returns the list of kinds ('warp' | 'copy') of its output frames."""
        c = self.calc
        thr = DEFAULT_SCENE_CHANGE_THRESHOLD if scene_change_threshold is None else scene_change_threshold
        host = NativeFilter(source_frame_time, target_frame_time, frame_output, thr)
        c.m_frameCount = 0              # a chunk starts like a new segment (HopperRender.cpp:840)
        kinds = []
        issued = drained = 0            # output frames handed to the D2H stream / to the sink
        base = c.downloadsIssued()      # the context's download counter at entry (a runner may run several chunks)
        R = len(self.outs)

        def drain(upto):
            nonlocal drained
            while drained < upto:
                c.waitDownload(base + drained)
                sink(drained, self.outs[drained % R].array, kinds[drained])
                drained += 1

        for k in range(chunk.first_frame, chunk.first_frame + chunk.n_frames):
            slot = self.ins[(k - chunk.first_frame) % len(self.ins)]
            fill(k, slot.array)                         # (the upload that last used this slot finished before an earlier hf_wait_flow)
            c.updateFrameAsync(slot)
            self.bytes_in += c.input_frame_bytes
            count = k + 1                               # the sequential run's m_frameCount at this frame
            if c.m_frameCount >= 3:
                c.calculateOpticalFlow()
                c.waitFlow()                            # m_totalFrameDelta of this period; side streams keep running
                host.push(count, c.m_totalFrameDelta)
            else:
                c.sync()                                # the first two frames of a segment: no chain to wait for (the input slot is reused three frames on)
            if k < chunk.first_period:
                continue                                # warm-up: ring, previous flow, delta history -- no output
            for t in chunk.scalars[k - chunk.first_period]:
                cut = host.detect(count)
                if count >= 3 and not cut:
                    c.warpFrames(t, frame_output); kinds.append("warp")
                else:
                    c.copyFrame(); kinds.append("copy")
                drain(issued - R + 1)                   # the slot about to be overwritten must have gone to the sink
                c.downloadFrameAsync(self.outs[issued % R])
                issued += 1
                self.bytes_out += c.output_frame_bytes
        drain(issued)
        c.sync()
        host.close()
        return kinds

    def close(self):
        self.calc.close()
        for p in self.ins + self.outs:
            p.free()
