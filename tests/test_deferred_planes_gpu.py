"""GPU: deferred phase planes (hf_batch_run_period, include/hopperflow.h).  Where the batched period warp is the workgroup-staged
kernel, a period only samples the grid of its new frame and the NEXT period's warp launch -- enqueued ahead of that period's
chain -- builds the full plane of the frame it reads anyway (hf_kernels.hip emit_plane_rows) instead of the stand-alone plane
kernel (hf_flow.hip prep_phase_fast_kernel, the re-laid frame that replaces calcDeltaSumsKernelSDR.h:78-100's strided sampling).
Everything observable must be bit-identical to the eager order: planes, flows, m_totalFrameDelta, output frames; and every period
that cannot defer (no outputs, diagnostic mode, a separate flow call) must fall back to the plane kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _upload(frames):
    from hopperrender_amd.calc import DeviceBuffer
    out = []
    for f in frames:
        b = DeviceBuffer(f.nbytes)
        b.upload(f)
        out.append(b)
    return out


def _run(cls, H, W, flags, frames_dev, n, schedule, ts_by_member, dt, R=16, check=None):
    """One batch of n members through hf_batch_run_period following `schedule` = [(with_outputs, mode, separate_calls)] per frame;
    returns per period: planes (slot 1) + completeness, blurred flows, deltas, outputs."""
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch
    members = [cls(H, W, search_radius=R, flags=flags) for _ in range(n)]
    batch = FlowBatch(members)
    n_ts = max(len(t) for t in ts_by_member)
    outs = [[DeviceBuffer(members[0].output_frame_bytes) for _ in range(n_ts)] for _ in range(n)]
    log = []
    for k, (with_out, mode, separate) in enumerate(schedule):
        ptrs = [frames_dev[(k + i) % len(frames_dev)].ptr for i in range(n)]     # members see the clip at different offsets
        flow = k >= 1
        if separate:
            batch.updateFramesDeviceRef(ptrs)
            if flow:
                batch.calculateOpticalFlow()
            if with_out:
                batch.interpolatePeriod(ts_by_member, [[b.ptr for b in o] for o in outs], mode)
        else:
            batch.runPeriod(batch.preparePeriod(ptrs, ts_by_member if with_out else None, [[b.ptr for b in o] for o in outs] if with_out else None,
                                                mode, calculate_flow=flow))
        rec = {"planes": [], "flows": [], "delta": [], "outs": []}
        for i, m in enumerate(members):
            m.sync()
            if check is not None and i not in check:
                for key in rec:
                    rec[key].append(None)
                continue
            rec["planes"].append(m.readPhasePlane(1))
            rec["flows"].append(m.readBlurredFlow(1).copy())
            rec["delta"].append(m.m_totalFrameDelta)
            rec["outs"].append([outs[i][j].download(dt).copy() for j in range(len(ts_by_member[i]))] if with_out else [])
        log.append(rec)
    batch.close()
    for m in members:
        m.close()
    for o in outs:
        for b in o:
            b.free()
    return log


@pytest.mark.parametrize("hdr,H,W,n", [(1, 2160, 3840, 4), (0, 4320, 7680, 2), (1, 4320, 7680, 1)])
def test_deferred_planes_equal_eager_planes(native_lib, hdr, H, W, n):
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    dt = np.uint16 if hdr else np.uint8
    sc = synth.Scene(H, W, bool(hdr), 77)
    frames = [sc.frame(k) for k in range(5)]
    dev = _upload(frames)
    ts = [[0.0, 0.1988, 0.3996, 0.5984, 0.7992][: 5 - (i % 2)] for i in range(n)]
    #            outputs, mode, separate calls
    schedule = [(False, 2, False),      # first frame: update only
                (False, 2, False),      # second: flow, no outputs -> the older frame's plane comes from the plane kernel (fallback)
                (True, 2, False),       # deferred: warp first, builds the plane
                (True, 2, False),
                (True, 3, False),       # diagnostic mode: not the staged kernel -> fallback
                (True, 2, False),
                (True, 0, True),        # the three calls: eager
                (True, 1, False),       # back to deferred, mode 1
                (True, 2, False)]
    base = capi.HF_FLAG_ASYNC
    eager = _run(cls, H, W, base | capi.HF_FLAG_BATCH_EAGER_PLANES, dev, n, schedule, ts, dt)
    lazy = _run(cls, H, W, base, dev, n, schedule, ts, dt)
    saw_incomplete = False
    for k, (a, b) in enumerate(zip(eager, lazy)):
        for i in range(n):
            pa, ca = a["planes"][i]
            pb, cb = b["planes"][i]
            assert ca, (k, i)                                   # eager planes are always complete
            if k >= 1:
                assert cb, (k, i)                               # after a period with a chain the older frame's plane is complete ...
                assert np.array_equal(pa, pb), (k, i)           # ... and equal to the plane kernel's, margins included
            else:
                saw_incomplete = saw_incomplete or not cb
            assert np.array_equal(a["flows"][i], b["flows"][i]), (k, i)
            assert a["delta"][i] == b["delta"][i], (k, i)
            assert len(a["outs"][i]) == len(b["outs"][i])
            for j, (x, y) in enumerate(zip(a["outs"][i], b["outs"][i])):
                assert np.array_equal(x, y), (k, i, j)
    for d in dev:
        d.free()


def test_deferral_is_taken_and_reported(native_lib):  # noqa
    """The deferred path is really taken at 2160p HDR (grid samples only after the update; the plane completes through the warp
    launch, not the plane kernel: checked by counting plane-kernel launches is not possible from here, so the completeness flag of
    ring slot 2 is the witness) and never at 1080p SDR (8-byte threads: the batch does not defer)."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR
    for cls, hdr, H, W, expect in ((OpticalFlowCalcHDR, True, 2160, 3840, True), (OpticalFlowCalcSDR, False, 1080, 1920, False)):
        sc = synth.Scene(H, W, hdr, 5)
        dev = _upload([sc.frame(k) for k in range(2)])
        members = [cls(H, W, search_radius=8, flags=capi.HF_FLAG_ASYNC) for _ in range(4)]
        batch = FlowBatch(members)
        assert batch.defersPlanes() == expect, (H, W)
        batch.runPeriod(batch.preparePeriod([dev[0].ptr] * 4, None, None, calculate_flow=False))
        batch.sync()
        _, complete = members[0].readPhasePlane(2)
        assert complete == (not expect), (H, W)
        batch.close()
        for m in members:
            m.close()
        for d in dev:
            d.free()


def test_deferred_planes_in_a_batch_of_20(native_lib):
    """20 members = a warp launch of 16 (staged kernel, builds the planes) + one of 4 (too few waves for the staged kernel: those
    members' planes come from the plane kernel): both kinds in one batch, same results as the eager batch."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import OpticalFlowCalcHDR
    H, W, n = 2160, 3840, 20
    sc = synth.Scene(H, W, True, 31)
    dev = _upload([sc.frame(k) for k in range(4)])
    ts = [[0.0, 0.1988, 0.3996, 0.5984, 0.7992] for _ in range(n)]
    schedule = [(False, 2, False), (True, 2, False), (True, 2, False), (True, 2, False)]
    check = (0, 15, 16, 19)
    eager = _run(OpticalFlowCalcHDR, H, W, capi.HF_FLAG_ASYNC | capi.HF_FLAG_BATCH_EAGER_PLANES, dev, n, schedule, ts, np.uint16, R=8, check=check)
    lazy = _run(OpticalFlowCalcHDR, H, W, capi.HF_FLAG_ASYNC, dev, n, schedule, ts, np.uint16, R=8, check=check)
    for k, (a, b) in enumerate(zip(eager, lazy)):
        for i in check:
            if k >= 1:
                assert b["planes"][i][1], (k, i)
                assert np.array_equal(a["planes"][i][0], b["planes"][i][0]), (k, i)
            assert np.array_equal(a["flows"][i], b["flows"][i]), (k, i)
            assert a["delta"][i] == b["delta"][i], (k, i)
            for j, (x, y) in enumerate(zip(a["outs"][i], b["outs"][i])):
                assert np.array_equal(x, y), (k, i, j)
    for d in dev:
        d.free()


def test_error_behaviour_is_that_of_the_three_calls(native_lib):
    """A period with an invalid blending scalar (> 1: opticalFlowCalcSDR.cpp:143-146) fails in hf_batch_run_period AFTER its update and
    its chain have been enqueued -- as with the three separate calls -- also on a batch that defers its planes and would issue the
    warps first: the flow of the failed period exists, and the following period equals the eager batch's."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR
    H, W, n = 2160, 3840, 4
    sc = synth.Scene(H, W, True, 13)
    dev = _upload([sc.frame(k) for k in range(4)])
    good, bad = [0.0, 0.25, 0.5, 0.75], [0.0, 0.25, 1.5, 0.75]
    results = []
    for flags in (capi.HF_FLAG_ASYNC | capi.HF_FLAG_BATCH_EAGER_PLANES, capi.HF_FLAG_ASYNC):
        members = [OpticalFlowCalcHDR(H, W, search_radius=8, flags=flags) for _ in range(n)]
        batch = FlowBatch(members)
        outs = [[DeviceBuffer(members[0].output_frame_bytes) for _ in good] for _ in range(n)]
        optr = [[b.ptr for b in o] for o in outs]
        batch.runPeriod(batch.preparePeriod([dev[0].ptr] * n, None, None, calculate_flow=False))
        batch.runPeriod(batch.preparePeriod([dev[1].ptr] * n, [good] * n, optr, 2))
        with pytest.raises(capi.HopperFlowError):
            batch.runPeriod(batch.preparePeriod([dev[2].ptr] * n, [good] * (n - 1) + [bad], optr, 2))
        batch.sync()
        flow_after_error = [m.readBlurredFlow(1).copy() for m in members]
        batch.runPeriod(batch.preparePeriod([dev[3].ptr] * n, [good] * n, optr, 2))
        batch.sync()
        results.append((flow_after_error, [m.readBlurredFlow(1).copy() for m in members], [m.m_totalFrameDelta for m in members],
                        [[b.download(np.uint16).copy() for b in o] for o in outs]))
        batch.close()
        for m in members:
            m.close()
        for o in outs:
            for b in o:
                b.free()
    (fe_a, f_a, d_a, o_a), (fe_b, f_b, d_b, o_b) = results
    assert d_a == d_b
    for i in range(n):
        assert fe_a[i].any(), i                                  # the failed period's chain did run
        assert np.array_equal(fe_a[i], fe_b[i]) and np.array_equal(f_a[i], f_b[i]), i
        for x, y in zip(o_a[i], o_b[i]):
            assert np.array_equal(x, y), i
    for d in dev:
        d.free()


def test_a_refused_flow_calculation_leaves_the_outputs_untouched(native_lib):
    """Members whose flow parameters differ make hf_batch_calculate_optical_flow fail (hf_batch.hip: "members differ ...").  With the three
    separate calls nothing is warped in that period; a plane-deferring hf_batch_run_period issues its warps BEFORE the chain, so it has
    to make the chain's argument checks first: the caller's output buffers keep their contents."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR
    H, W, n = 2160, 3840, 4
    sc = synth.Scene(H, W, True, 21)
    dev = _upload([sc.frame(k) for k in range(3)])
    ts = [0.0, 0.25, 0.5, 0.75]
    members = [OpticalFlowCalcHDR(H, W, search_radius=8, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    batch = FlowBatch(members)
    assert batch.defersPlanes()
    outs = [[DeviceBuffer(members[0].output_frame_bytes) for _ in ts] for _ in range(n)]
    marker = np.full(members[0].output_frame_bytes // 2, 0x5A5A, np.uint16)
    for o in outs:
        for b in o:
            b.upload(marker)
    optr = [[b.ptr for b in o] for o in outs]
    batch.runPeriod(batch.preparePeriod([dev[0].ptr] * n, None, None, calculate_flow=False))
    batch.runPeriod(batch.preparePeriod([dev[1].ptr] * n, None, None))
    members[2].m_opticalFlowSearchRadius = 7                      # the governor of ONE member moved: the batch cannot run one chain for all
    with pytest.raises(capi.HopperFlowError):
        batch.runPeriod(batch.preparePeriod([dev[2].ptr] * n, [ts] * n, optr, 2))
    batch.sync()
    for i, o in enumerate(outs):
        for j, b in enumerate(o):
            assert np.array_equal(b.download(np.uint16), marker), (i, j)
    members[2].m_opticalFlowSearchRadius = 8                      # and the batch is usable again
    batch.runPeriod(batch.preparePeriod(None, [ts] * n, optr, 2))
    batch.sync()
    assert not np.array_equal(outs[0][1].download(np.uint16), marker)
    batch.close()
    for m in members:
        m.close()
    for b in dev + [x for o in outs for x in o]:
        b.free()
