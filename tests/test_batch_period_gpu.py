"""GPU: the batched period calls of hf_batch -- hf_batch_update_frames_device_ref (the phase planes of every member's
new frame in one launch) and hf_batch_interpolate_period (the warps of a source period of EVERY member in one launch)
-- must give each member exactly what its own hf_update_frame_device_ref / hf_interpolate_period_ex give."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hdr,H,W,n,R,mode", [(0, 360, 640, 3, 8, 2), (1, 360, 640, 8, 16, 2), (0, 1080, 1920, 4, 16, 2),
                                             (1, 2160, 3840, 2, 5, 2), (0, 274, 486, 5, 16, 2), (1, 360, 640, 4, 9, 0),
                                             (0, 360, 640, 3, 9, 1), (0, 360, 640, 2, 7, 4), (1, 338, 600, 3, 6, 2), (0, 180, 320, 16, 9, 2), (1, 180, 320, 13, 16, 2),
                                             (0, 180, 320, 32, 16, 2), (1, 270, 480, 27, 11, 2)])
def test_batched_period_equals_single_contexts(native_lib, hdr, H, W, n, R, mode):
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from hopperrender_amd.protocol import SOURCE_24, TARGET_120, BlendSchedule
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    scenes = [synth.Scene(H, W, bool(hdr), 300 + 11 * i) for i in range(n)]
    frames = [[sc.frame(k) for k in range(6)] for sc in scenes]
    dev = [[DeviceBuffer(f.nbytes) for f in fs] for fs in frames]
    for i in range(n):
        for k in range(6):
            dev[i][k].upload(frames[i][k])
    singles = [cls(H, W, search_radius=R, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    members = [cls(H, W, search_radius=R, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    batch = FlowBatch(members)
    # members run different phases of the 24 -> 120 schedule: 5 or 6 outputs each, different scalars
    plans = [BlendSchedule(SOURCE_24, TARGET_120).plan(10 + i)[i:] for i in range(n)]
    K = 6
    outs_s = [[DeviceBuffer(singles[0].output_frame_bytes) for _ in range(K)] for _ in range(n)]
    outs_b = [[DeviceBuffer(singles[0].output_frame_bytes) for _ in range(K)] for _ in range(n)]
    dt = np.uint16 if hdr else np.uint8
    for k in range(6):
        for i in range(n):
            singles[i].updateFrameDeviceRef(dev[i][k].ptr)
        batch.updateFramesDeviceRef([dev[i][k].ptr for i in range(n)])
        if k < 2:
            continue
        for i in range(n):
            singles[i].calculateOpticalFlow()
        batch.calculateOpticalFlow()
        ts = [plans[i][k] for i in range(n)]
        for i in range(n):
            singles[i].interpolateOnly(ts[i], [b.ptr for b in outs_s[i]], mode)
        batch.interpolatePeriod(ts, [[b.ptr for b in outs_b[i]] for i in range(n)], mode)
        for i in range(n):
            singles[i].sync(); members[i].sync()
            assert members[i].m_frameCount == singles[i].m_frameCount == k + 1
            assert members[i].m_totalFrameDelta == singles[i].m_totalFrameDelta, (k, i)
            assert (members[i].readBlurredFlow(1) == singles[i].readBlurredFlow(1)).all(), (k, i)
            for j in range(len(ts[i])):
                assert (outs_s[i][j].download(dt) == outs_b[i][j].download(dt)).all(), (k, i, j)
    batch.close()
    for c in singles + members:
        c.close()


@pytest.mark.parametrize("hdr,H,W,n,R,mode", [(1, 2160, 3840, 8, 16, 2), (1, 2160, 3840, 16, 9, 2), (0, 2160, 3840, 6, 16, 2), (1, 2154, 3832, 8, 12, 2),
                                             (1, 2160, 3840, 8, 16, 0), (0, 2160, 3840, 8, 11, 1), (1, 1080, 1920, 12, 16, 2), (0, 1082, 1924, 16, 16, 2),
                                             (1, 2160, 3840, 3, 16, 2), (1, 2160, 3840, 32, 16, 2), (0, 1080, 1920, 32, 16, 2)])
def test_large_batched_periods_equal_single_contexts(native_lib, hdr, H, W, n, R, mode):
    """Batched periods at the sizes the bench runs (up to 16 members of 2160p / 1080p frames in one launch: the unit decoding
    (member, tile block, chunk) of warp_fast_kernel and the XCD banding only get large here).  A member's outputs must equal
    those of a single context, which is checked against the oracle and the golden vectors elsewhere.  Includes frames
    whose planes end inside a wave tile (2154 rows = 269 tiles + 2 rows, 1077 chroma rows; 1924 columns)."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from hopperrender_amd.protocol import SOURCE_24, TARGET_120, BlendSchedule
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    scenes = [synth.Scene(H, W, bool(hdr), 700 + 13 * i) for i in range(2)]
    frames = [[sc.frame(k) for k in range(4)] for sc in scenes]
    dev = [[DeviceBuffer(f.nbytes) for f in fs] for fs in frames]
    for i in range(2):
        for k in range(4):
            dev[i][k].upload(frames[i][k])
    src = lambda i, k: dev[i % 2][(k + i // 2) % 4]          # member i: scene i % 2, started i // 2 frames in
    singles = [cls(H, W, search_radius=R, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    members = [cls(H, W, search_radius=R, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    batch = FlowBatch(members)
    plans = [BlendSchedule(SOURCE_24, TARGET_120).plan(10 + i)[i:] for i in range(n)]
    outs_s = [DeviceBuffer(singles[0].output_frame_bytes) for _ in range(6)]
    outs_b = [[DeviceBuffer(singles[0].output_frame_bytes) for _ in range(6)] for _ in range(n)]
    dt = np.uint16 if hdr else np.uint8
    for k in range(4):
        for i in range(n):
            singles[i].updateFrameDeviceRef(src(i, k).ptr)
        batch.updateFramesDeviceRef([src(i, k).ptr for i in range(n)])
        if k < 2:
            continue
        for i in range(n):
            singles[i].calculateOpticalFlow()
        batch.calculateOpticalFlow()
        ts = [plans[i][k] for i in range(n)]
        batch.interpolatePeriod(ts, [[b.ptr for b in outs_b[i]] for i in range(n)], mode)
        for i in range(n):
            singles[i].interpolateOnly(ts[i], [b.ptr for b in outs_s], mode)
            singles[i].sync(); members[i].sync()
            for j in range(len(ts[i])):
                assert (outs_s[j].download(dt) == outs_b[i][j].download(dt)).all(), (k, i, j)
    batch.close()
    for c in singles + members:
        c.close()


def test_run_period_is_the_three_calls(native_lib):
    """hf_batch_run_period (one native call per batch and source period, what bench.py and examples/hf_batch_driver.c issue)
    == hf_batch_update_frames_device_ref + hf_batch_calculate_optical_flow + hf_batch_interpolate_period, also with parts
    skipped; hf_batch_sync == hf_sync of every member."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR
    H, W, n, R = 360, 640, 5, 13
    scenes = [synth.Scene(H, W, True, 40 + i) for i in range(n)]
    dev = []
    for sc in scenes:
        row = []
        for k in range(6):
            f = sc.frame(k)
            b = DeviceBuffer(f.nbytes); b.upload(f); row.append(b)
        dev.append(row)
    mk = lambda: [OpticalFlowCalcHDR(H, W, search_radius=R, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    three, one = mk(), mk()
    b3, b1 = FlowBatch(three), FlowBatch(one)
    ts = [[0.0, 0.25, 0.5, 0.75, 0.9][:3 + i % 3] for i in range(n)]
    o3 = [[DeviceBuffer(three[0].output_frame_bytes) for _ in range(5)] for _ in range(n)]
    o1 = [[DeviceBuffer(three[0].output_frame_bytes) for _ in range(5)] for _ in range(n)]
    for k in range(6):
        ptrs = [dev[i][k].ptr for i in range(n)]
        b3.updateFramesDeviceRef(ptrs)
        if k < 2:
            b1.runPeriod(b1.preparePeriod(ptrs, None, None, calculate_flow=False))      # update only
            continue
        b3.calculateOpticalFlow()
        b3.interpolatePeriod(ts, [[x.ptr for x in o3[i]] for i in range(n)], 2)
        if k == 3:    # the same period in two calls: update + flow, then the warps alone
            b1.runPeriod(b1.preparePeriod(ptrs, None, None))
            b1.runPeriod(b1.preparePeriod(None, ts, [[x.ptr for x in o1[i]] for i in range(n)], 2, calculate_flow=False))
        else:
            b1.runPeriod(b1.preparePeriod(ptrs, ts, [[x.ptr for x in o1[i]] for i in range(n)], 2))
        b1.sync()
        for i in range(n):
            three[i].sync()
            assert one[i].m_frameCount == three[i].m_frameCount == k + 1
            assert one[i].m_totalFrameDelta == three[i].m_totalFrameDelta
            for j in range(len(ts[i])):
                assert (o1[i][j].download(np.uint16) == o3[i][j].download(np.uint16)).all(), (k, i, j)
    with pytest.raises(capi.HopperFlowError) as e:      # same error behaviour as the calls it stands for
        b1.runPeriod(b1.preparePeriod(None, [[1.5]] * n, [[o1[i][0].ptr] for i in range(n)], 2, calculate_flow=False))
    assert e.value.code == capi.HF_ERR_INVALID_ARGUMENT and "greater than 1.0" in str(e.value)
    with pytest.raises(capi.HopperFlowError) as e:      # a NULL frame is caught before ANY member's ring is touched
        b1.runPeriod(b1.preparePeriod([dev[0][0].ptr] * (n - 1) + [0], None, None, calculate_flow=False))
    assert e.value.code == capi.HF_ERR_INVALID_ARGUMENT
    assert all(m.m_frameCount == 6 for m in one)
    b1.runPeriod(b1.preparePeriod(None, ts, [[x.ptr for x in o1[i]] for i in range(n)], 2, calculate_flow=False))
    b1.sync()
    for i in range(n):     # ... so the members still warp the same frames
        for j in range(len(ts[i])):
            assert (o1[i][j].download(np.uint16) == o3[i][j].download(np.uint16)).all(), ("after the failed update", i, j)
    b3.close(); b1.close()
    for c in three + one:
        c.close()


def test_first_suitable_device_on_this_box(native_lib):
    """device_index = -1 (what the C++ drop-in passes): detectDevices settles on the first suitable HIP device."""
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    c = OpticalFlowCalcSDR(180, 320, device_index=-1)
    assert c.device_index == 0 and native_lib.hf_get_device(c._ctx) == 0
    c.close()


@pytest.mark.parametrize("H,W,mode", [(180, 320, 2), (360, 640, 2), (360, 640, 0), (360, 640, 1)])
def test_batched_period_matches_oracle(native_lib, H, W, mode):
    """... and through the oracle the reference: one batched period against the CPU restatement -- at 180p (rs = 0: one-pixel flow cells,
    the generic warp kernel, one launch per output) and at 360p (rs = 1: two-pixel = two-byte cells of an 8-bit frame, which the fused
    period warp takes since round 6: runs read with plain element loads, four cells per 8-byte thread)."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcSDR
    from oracle import oracle
    R, n = 8, 3
    g = oracle.make_geom(0, H, W)
    scenes = [synth.Scene(H, W, False, 900 + i) for i in range(n)]
    frames = [[sc.frame(k) for k in range(4)] for sc in scenes]
    dev = [[DeviceBuffer(f.nbytes) for f in fs] for fs in frames]
    for i in range(n):
        for k in range(4):
            dev[i][k].upload(frames[i][k])
    members = [OpticalFlowCalcSDR(H, W, search_radius=R, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    batch = FlowBatch(members)
    outs = [[DeviceBuffer(members[0].output_frame_bytes) for _ in range(3)] for _ in range(n)]
    ts = [[0.0, 0.3996, 0.7992], [0.1988, 0.5984, 0.998], [0.5]]
    for k in range(4):
        batch.updateFramesDeviceRef([dev[i][k].ptr for i in range(n)])
        if k >= 2:
            batch.calculateOpticalFlow()
    batch.interpolatePeriod(ts, [[b.ptr for b in outs[i]] for i in range(n)], mode)
    for i in range(n):
        members[i].sync()
        _, blur, tot, oob = oracle.calculate_optical_flow(frames[i][1], frames[i][2], g, R)
        assert oob == 0
        assert (members[i].readBlurredFlow(0) == blur).all()
        for j, t in enumerate(ts[i]):
            ref = oracle.warp_frames(frames[i][1], frames[i][2], blur, g, np.float32(t), mode)
            assert (outs[i][j].download(np.uint8) == ref).all(), (i, j)
    batch.close()
    for c in members:
        c.close()


def test_batch_state_errors(native_lib):
    """HF_ERR_STATE: a context joins one batch at a time; asynchronous host I/O is not available to batch members.
    A failed hf_batch_create leaves every context as it was."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import FlowBatch, OpticalFlowCalcSDR, PinnedArray
    a, b, c = (OpticalFlowCalcSDR(180, 320, flags=capi.HF_FLAG_ASYNC) for _ in range(3))
    odd = OpticalFlowCalcSDR(184, 320, flags=capi.HF_FLAG_ASYNC)
    ba = FlowBatch([a, b])
    with pytest.raises(capi.HopperFlowError) as e:
        FlowBatch([c, a])
    assert e.value.code == capi.HF_ERR_STATE and "already belongs to a batch" in str(e.value)
    with pytest.raises(capi.HopperFlowError) as e:
        FlowBatch([c, odd])
    assert e.value.code == capi.HF_ERR_INVALID_ARGUMENT
    many = [OpticalFlowCalcSDR(180, 320, flags=capi.HF_FLAG_ASYNC) for _ in range(33)]
    with pytest.raises(capi.HopperFlowError) as e:
        FlowBatch(many)          # a launch carries the buffer pointers of at most 32 members
    assert e.value.code == capi.HF_ERR_INVALID_ARGUMENT and "at most 32" in str(e.value)
    FlowBatch(many[:32]).close()
    for x in many:
        x.close()
    pin = PinnedArray(a.input_frame_bytes, np.uint8)
    with pytest.raises(capi.HopperFlowError) as e:
        a.updateFrameAsync(pin)
    assert e.value.code == capi.HF_ERR_STATE
    f = synth.Scene(180, 320, False, 3).frame(0)
    for _ in range(3):
        c.updateFrame(f)          # c was named in two failed creates: still an ordinary, working context
    c.calculateOpticalFlow(); c.sync()
    assert c.m_frameCount == 3 and (c.readBlurredFlow(1) == 0).all()     # identical frames: zero flow
    ba.close()
    FlowBatch([c, a]).close()     # and free to join a batch once the other one is gone
    pin.free()
    for x in (a, b, c, odd):
        x.close()
