"""GPU: the fused period warp in exactly the launch shapes bench.py times, judged by the REFERENCE (golden vectors
of tests/golden/make_golden.py) and by the pinned oracle -- not by another HIP launch.

bench.py's default workload runs `warp_fast_kernel<unsigned short, 8, 2, 2, 16, true>` with ALL outputs of a source period
per thread (`out_chunk = 6`, hf_kernels.hip launch_warp_fast) and, batched, 16-wave workgroups; frames up to 1080p take
that path only inside a batch of >= 11 members (a single context produces one output per thread there).  These tests run
  (a) one context, 2160p HDR and 1080p SDR, five outputs per period, modes 0 / 1 / 2;
  (b) two batches of 16 members side by side on their two batch streams (the bench's operating point), members fed the
      golden frames with different 5- and 6-output schedules;
and compare every output frame whose blending scalar the golden file covers by SHA-256 with the reference's frame
(reference warpFrameKernel{SDR,HDR}.h:116-184 run through oracle/_ref on an MI355X), every other one with the oracle
evaluated on the reference's own blurred flow."""
import numpy as np
import pytest

from helpers import Golden, sha

pytestmark = pytest.mark.gpu

T5 = [0.1988, 0.3996, 0.5984, 0.7992, 0.998]     # 24 -> 60 fps scalars (HopperRender.cpp:1192-1197); the golden files hold 0.3996 and 0.7992
T6 = T5 + [0.0]


class Judge:
    """Expected output frames of (mode, t) for one golden key: the reference's SHA where the golden file has the frame,
    otherwise the oracle on the reference's blurred flow (computed once)."""

    def __init__(self, g, key, frames):
        from oracle import oracle
        self.g, self.key, self.frames, self.oracle = g, key, frames, oracle
        c = g.case
        self.geom = oracle.make_geom(c["hdr"], c["H"], c["W"], c["si"], c["so"], c.get("max_res", 270))
        self.flow = g.arr(key, "blur_a")            # flow f1 -> f2 as the reference computed it
        self.cache = {}
        self.n_golden = self.n_oracle = 0

    def check(self, out, mode, t, what):
        fname = f"warp_m{mode}_t{t}"
        k = (mode, t)
        if k not in self.cache:
            if fname in self.g.frame_names(self.key):
                assert sha(out) == self.g.frame_sha(self.key, fname), f"{what}: differs from the reference's frame {fname}"
                self.cache[k] = out.copy()
            else:
                self.cache[k] = self.oracle.warp_frames(self.frames[1], self.frames[2], self.flow, self.geom, np.float32(t), mode)
        if fname in self.g.frame_names(self.key):
            self.n_golden += 1
        else:
            self.n_oracle += 1
        assert np.array_equal(out, self.cache[k]), f"{what}: mode {mode} t {t} differs"


def _upload(frames):
    from hopperrender_amd.calc import DeviceBuffer
    dev = []
    for f in frames:
        b = DeviceBuffer(f.nbytes)
        b.upload(f)
        dev.append(b)
    return dev


def _make(case, R, delta, nb, flags):
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    cls = OpticalFlowCalcHDR if case["hdr"] else OpticalFlowCalcSDR
    return cls(case["H"], case["W"], case["si"], case["so"], delta, nb, 0.0, 255.0, case.get("max_res", 270), search_radius=R, flags=flags)


@pytest.mark.parametrize("name", ["hdr_2160p", "sdr_1080p"])
def test_fused_period_of_one_context_matches_the_reference(native_lib, name):
    from hopperrender_amd import capi
    from hopperrender_amd.calc import DeviceBuffer
    g = Golden(name)
    frames = g.frames()
    dev = _upload(frames)
    dt = np.uint16 if g.case["hdr"] else np.uint8
    for key in g.keys:
        R, delta, nb = g.params(key)
        judge = Judge(g, key, frames)
        c = _make(g.case, R, delta, nb, capi.HF_FLAG_ASYNC)
        outs = [DeviceBuffer(c.output_frame_bytes) for _ in range(6)]
        for k in range(4):
            c.updateFrameDeviceRef(dev[k].ptr)
            if k >= 2:
                c.calculateOpticalFlow()
        c.sync()
        assert (c.readBlurredFlow(0) == judge.flow).all()
        for mode in (2, 0, 1):
            for ts in (T5, T6[::-1]):
                c.interpolateOnly(ts, [b.ptr for b in outs], mode)
                c.sync()
                for j, t in enumerate(ts):
                    judge.check(outs[j].download(dt), mode, t, f"{name} {key} single fused period")
        assert judge.n_golden >= 12 and judge.n_oracle >= 18
        c.close()
        for b in outs:
            b.free()


@pytest.mark.parametrize("name,modes", [("hdr_2160p", (2, 0)), ("sdr_1080p", (2, 1))])
def test_two_batches_of_16_match_the_reference(native_lib, name, modes):
    """bench.py's operating point: 32 pair streams = 2 batch streams of 16, one fused warp launch per batch and period
    (16-wave workgroups at 2160p; at 1080p the batch is what switches the launch to all outputs per thread)."""
    from hopperrender_amd import capi
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch
    g = Golden(name)
    frames = g.frames()
    dev = _upload(frames)
    dt = np.uint16 if g.case["hdr"] else np.uint8
    key = "R16_d8_n6"
    R, delta, nb = g.params(key)
    judge = Judge(g, key, frames)
    n, nb_batches = 16, 2
    members = [_make(g.case, R, delta, nb, capi.HF_FLAG_ASYNC | capi.HF_FLAG_NO_TIMING) for _ in range(n * nb_batches)]
    batches = [FlowBatch(members[b * n:(b + 1) * n]) for b in range(nb_batches)]
    outs = [[DeviceBuffer(members[0].output_frame_bytes) for _ in range(6)] for _ in members]
    # every member: its own rotation of the schedule, 5 or 6 outputs (the kernel's per-member n_out and scalars)
    plans = [(T6[i % 6:] + T6[:i % 6])[:5 + (i % 2)] for i in range(len(members))]
    for k in range(4):
        for b in batches:
            b.updateFramesDeviceRef([dev[k].ptr] * n)
            if k >= 2:
                b.calculateOpticalFlow()
    for mode in modes:
        for bi, b in enumerate(batches):      # both batches in flight, nothing in between (as in bench.py)
            lo = bi * n
            b.interpolatePeriod(plans[lo:lo + n], [[x.ptr for x in outs[lo + i]] for i in range(n)], mode)
        for i, m in enumerate(members):
            m.sync()
            if mode == modes[0]:
                assert (m.readBlurredFlow(0) == judge.flow).all(), i
            for j, t in enumerate(plans[i]):
                judge.check(outs[i][j].download(dt), mode, t, f"{name} batch member {i}")
    assert judge.n_golden >= 2 * 2 * 16 and judge.n_oracle > judge.n_golden
    for b in batches:
        b.close()
    for m in members:
        m.close()
    for bufs in outs:
        for b in bufs:
            b.free()


@pytest.mark.parametrize("hdr,H,W,n", [(0, 2160, 3840, 1), (0, 2160, 3840, 3), (1, 1080, 1920, 12), (0, 1440, 2560, 4), (0, 4320, 7680, 2), (1, 4320, 7680, 2)])
def test_fused_period_of_other_launch_shapes_matches_the_oracle(native_lib, hdr, H, W, n):
    """The fused period in the launch shapes the golden files do not reach, judged by the pinned oracle at full size: 8-bit 2160p
    (16-byte threads with TWO flow cells each: the merged-run path), 16-bit 1080p in a batch of 12 (all outputs per thread with two
    cells per thread), 8-bit 1440p (rs = 3, 16 elements per thread), and 4320p in batches of 2 (rs = 4: a 16-byte thread lies inside ONE
    flow cell also for 8-bit frames, two threads per cell for 16-bit ones -- the unsigned-char instantiation and the two-lanes-per-cell
    case of the LDS-staged workgroup kernel)."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    R = 16
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    dt = np.uint16 if hdr else np.uint8
    g = oracle.make_geom(hdr, H, W)
    sc = synth.Scene(H, W, bool(hdr), 4242)
    frames = [sc.frame(k) for k in range(4)]
    dev = _upload(frames)
    _, flow, tot, oob = oracle.calculate_optical_flow(frames[1], frames[2], g, R)
    assert oob == 0
    ts = [0.0, 0.1988, 0.3996, 0.5984, 0.7992, 0.998]
    want = {t: oracle.warp_frames(frames[1], frames[2], flow, g, np.float32(t), 2) for t in ts}
    members = [cls(H, W, search_radius=R, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    outs = [[DeviceBuffer(members[0].output_frame_bytes) for _ in ts] for _ in members]
    plans = [(ts[i % 6:] + ts[:i % 6])[:6 - (i % 2)] for i in range(n)]
    if n == 1:
        c = members[0]
        for k in range(4):
            c.updateFrameDeviceRef(dev[k].ptr)
            if k >= 2:
                c.calculateOpticalFlow()
        c.interpolateOnly(plans[0], [b.ptr for b in outs[0]], 2)
    else:
        batch = FlowBatch(members)
        for k in range(4):
            batch.runPeriod(batch.preparePeriod([dev[k].ptr] * n, None, None, calculate_flow=k >= 2))
        batch.runPeriod(batch.preparePeriod(None, plans, [[b.ptr for b in o] for o in outs], 2, calculate_flow=False))
    for i, m in enumerate(members):
        m.sync()
        assert (m.readBlurredFlow(0) == flow).all(), i
        for j, t in enumerate(plans[i]):
            assert np.array_equal(outs[i][j].download(dt), want[t]), (i, j, t)
    if n > 1:
        batch.close()
    for m in members:
        m.close()


@pytest.mark.parametrize("flow_kind", ["uniform_small", "uniform_large", "diverging", "noise", "vertical_fast", "half_and_half"])
def test_staged_warp_and_its_fallbacks_match_oracle_and_global_path(native_lib, flow_kind):
    """The batched 2160p HDR period warp copies the window of each source frame that all outputs of a 128 x 32 workgroup tile read
    into LDS (warp_wg_kernel) and falls back to the global path per workgroup when the runs do not fit the window (fast or
    diverging motion), touch the mirror zone, or a wave is partial.  Injected flow fields force each case: members of a batch of 4
    (staged kernel) must equal a single context (global-path kernel) bit for bit, and the oracle on one output."""
    from hopperrender_amd import capi, synth
    from hopperrender_amd.calc import DeviceBuffer, FlowBatch, OpticalFlowCalcHDR
    from oracle import oracle
    H, W, n = 2160, 3840, 4
    g = oracle.make_geom(1, H, W)
    lw, lh = g.lw, g.lh
    rng = np.random.default_rng(7)
    flow = np.zeros((2, lh, lw), np.int16)
    if flow_kind == "uniform_small":
        flow[0], flow[1] = 9, -5                               # every window fits: all interior workgroups staged
    elif flow_kind == "uniform_large":
        flow[0], flow[1] = 230, -140                           # 0.8 x 140 rows of displacement range: no window fits -> global path
    elif flow_kind == "diverging":
        flow[0] = np.linspace(-200, 200, lw).astype(np.int16)[None, :]          # x displacement varies by 13 pixels per tile
        flow[1] = np.linspace(-90, 90, lh).astype(np.int16)[:, None]
    elif flow_kind == "noise":
        flow[:] = rng.integers(-40, 41, size=flow.shape)       # every lane of a wave differs
    elif flow_kind == "vertical_fast":
        flow[1] = 64                                           # 51 rows of range at t = 0.8: over the 12 KB window of a 32-row tile
    else:
        flow[0, :, : lw // 2], flow[1, :, : lw // 2] = 4, 2    # left half fits, right half does not: both paths in one launch
        flow[0, :, lw // 2:], flow[1, :, lw // 2:] = -300, 200
    sc = synth.Scene(H, W, True, 99)
    frames = [sc.frame(k) for k in range(3)]
    dev = _upload(frames)
    ts = [0.0, 0.1988, 0.5, 0.7992, 0.998]
    single = OpticalFlowCalcHDR(H, W, search_radius=5, flags=capi.HF_FLAG_ASYNC)
    members = [OpticalFlowCalcHDR(H, W, search_radius=5, flags=capi.HF_FLAG_ASYNC) for _ in range(n)]
    for c in [single] + members:
        for k in range(3):
            c.updateFrameDeviceRef(dev[k].ptr)
        c.sync()
        c.writeBlurredFlow(0, flow)
    outs_s = [DeviceBuffer(single.output_frame_bytes) for _ in ts]
    outs_b = [[DeviceBuffer(single.output_frame_bytes) for _ in ts] for _ in range(n)]
    batch = FlowBatch(members)
    plans = [ts[i:] + ts[:i] for i in range(n)]
    for mode in (2, 0, 1):
        single.interpolateOnly(ts, [b.ptr for b in outs_s], mode)
        single.sync()
        want = {t: outs_s[j].download(np.uint16) for j, t in enumerate(ts)}
        if mode == 2:
            t = ts[3]
            assert np.array_equal(want[t], oracle.warp_frames(frames[0], frames[1], flow, g, np.float32(t), 2)), flow_kind
        batch.interpolatePeriod(plans, [[b.ptr for b in o] for o in outs_b], mode)
        for i, m in enumerate(members):
            m.sync()
            for j, t in enumerate(plans[i]):
                assert np.array_equal(outs_b[i][j].download(np.uint16), want[t]), (flow_kind, mode, i, t)
    batch.close()
    for c in [single] + members:
        c.close()
