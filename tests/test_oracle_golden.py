"""CPU: the oracle (oracle/hf_oracle.c) against the committed golden vectors, which are outputs of the
reference itself (oracle/_ref on an MI355X, tests/golden/make_golden.py).  Bit-exact everywhere except the
diagnostic HSV mode (atan2/fmod of the OpenCL runtime): <= 2 LSB of 8 bits."""
import json
import os

import numpy as np
import pytest

from helpers import ALL_GOLDEN, GOLDEN_DIR, Golden, parse_frame_name, sha
from oracle import oracle

SMALL = [n for n in ALL_GOLDEN if n not in ("sdr_1080p", "hdr_2160p")]


@pytest.mark.parametrize("name", SMALL + ["sdr_1080p", "hdr_2160p"])
def test_flow_matches_reference(name):
    g = Golden(name)
    frames = g.frames()
    c = g.case
    geom = oracle.make_geom(c["hdr"], c["H"], c["W"], c["si"], c["so"], c.get("max_res", 270))
    keys = g.keys if name in SMALL else g.keys[-1:]          # full-size cases: one parameter set (CPU time)
    for key in keys:
        R, delta, nb = g.params(key)
        assert (geom.rs, geom.lw, geom.lh) == tuple(g.meta[key]["geom"][k] for k in ("rs", "lw", "lh"))
        off, blur, tot, oob = oracle.calculate_optical_flow(frames[1], frames[2], geom, R, 0, delta, nb, 4)
        assert oob == 0
        assert (off == g.arr(key, "off_a")).all()
        assert (blur == g.arr(key, "blur_a")).all()
        assert tot == g.meta[key]["stats_a"]["total_frame_delta"]
        if name in SMALL:
            off, blur, tot, _ = oracle.calculate_optical_flow(frames[2], frames[3], geom, R, 0, delta, nb, 4)
            assert (off == g.arr(key, "off_b")).all() and (blur == g.arr(key, "blur_b")).all()
            assert tot == g.meta[key]["stats_b"]["total_frame_delta"]


@pytest.mark.parametrize("name", SMALL + ["sdr_1080p"])
def test_warp_and_copy_match_reference(name):
    g = Golden(name)
    frames = g.frames()
    c = g.case
    geom = oracle.make_geom(c["hdr"], c["H"], c["W"], c["si"], c["so"], c.get("max_res", 270))
    key = g.keys[-1]
    flow = g.arr(key, "blur_a")
    for fname in g.frame_names(key):
        kind, mode, t, (bk, wh) = parse_frame_name(fname)
        if kind == "warp":
            out = oracle.warp_frames(frames[1], frames[2], flow, geom, t, mode, bk, wh)
        else:
            out = oracle.copy_frame(frames[1], geom, bk, wh)
        if mode == 3:
            if g.has(key, fname):
                d = np.abs(out.astype(np.int64) - g.arr(key, fname).astype(np.int64))
                assert d.max() <= (2 * 256 if c["hdr"] else 2)
            continue
        assert sha(out) == g.frame_sha(key, fname), f"{name} {key} {fname}"
        if g.has(key, fname):
            assert (out == g.arr(key, fname)).all()


def test_levels_ramp_matches_reference():
    """Every code value through copyFrame; the reference's OpenCL build divides through v_rcp_f32, which is
    1 ulp low for 180 and 46080 = 180*256 (the one setting below where the correctly rounded reciprocal differs)."""
    z = np.load(os.path.join(GOLDEN_DIR, "levels_ramp.npz"))
    H = W = 256
    try:
        for name in z.files:
            if name == "meta":
                continue
            kind, ph, b, w = name.split("_")
            hdr, phase, bk, wh = kind == "hdr", int(ph[1:]), float(b[1:]), float(w[1:])
            n = (H + H // 2) * W
            a = np.arange(n, dtype=np.uint32)
            a[H * W:] += phase * (n - H * W)
            f = (a % 65536).astype(np.uint16) if hdr else (a % 256).astype(np.uint8)
            rcp = None
            if wh == 180.0:   # v_rcp_f32(180 * 2^k) is 1 ulp below the correctly rounded reciprocal (measured on gfx950)
                y = np.float32(46080.0 if hdr else 180.0)
                rcp = {float(y): float(np.nextafter(np.float32(1.0) / y, np.float32(0.0)))}
            oracle.set_flavour(1, 1, rcp)
            out = oracle.copy_frame(f, oracle.make_geom(hdr, H, W), bk, wh)
            assert (out == z[name]).all(), name
    finally:
        oracle.set_flavour(1, 1, None)


def test_strict_ieee_flavour_is_within_one_lsb():
    """The strict-IEEE reading of the reference source (what an x86 build would do) stays within 1 LSB
    (8-bit) / 2 codes (16-bit) of the gfx950 OpenCL build -- the tolerance SURVEY.md section 8(c) states."""
    g = Golden("sdr_180p")
    frames = g.frames()
    geom = oracle.make_geom(0, 180, 320)
    key = "R16_d8_n6"
    try:
        oracle.set_flavour(0, 0)
        out = oracle.warp_frames(frames[1], frames[2], g.arr(key, "blur_a"), geom, 0.3996, 2)
    finally:
        oracle.set_flavour(1, 1)
    d = np.abs(out.astype(np.int64) - g.arr(key, "warp_m2_t0.3996").astype(np.int64))
    assert 0 < d.max() <= 1


def test_schedule_and_geometry_helpers():
    L = oracle.lib()
    assert L.hfo_initial_window(480, 270) == 256 and L.hfo_initial_window(320, 180) == 256
    assert L.hfo_initial_window(64, 36) == 32 and L.hfo_initial_window(256, 100) == 128
    assert L.hfo_iterations(256, 0) == 8 and L.hfo_iterations(256, 3) == 3 and L.hfo_iterations(256, 99) == 8
    assert [L.hfo_rel_offset(z, 16) for z in range(16)] == [-64, -49, -36, -25, -16, -9, -4, -1, 0, 1, 4, 9, 16, 25, 36, 49]
    assert [L.hfo_rel_offset(z, 5) for z in range(5)] == [-4, -1, 0, 1, 4]
    for (H, W, rs, lw, lh) in [(1080, 1920, 2, 480, 270), (2160, 3840, 3, 480, 270), (360, 640, 1, 320, 180), (338, 600, 1, 300, 169)]:
        g = oracle.make_geom(0, H, W)
        assert (g.rs, g.lw, g.lh) == (rs, lw, lh)


def test_flow_edge_cases():
    """Identical frames -> zero flow; uint32 wrap-around of the window sums is preserved."""
    from hopperrender_amd import synth
    f = synth.Scene(90, 160, False, 3).frame(0)
    g = oracle.make_geom(0, 90, 160)
    off, blur, tot, _ = oracle.calculate_optical_flow(f, f.copy(), g, 16)
    assert not off.any() and not blur.any()
    assert tot > 0   # bug-compatible m_totalFrameDelta: read at layer R/2-1 (offset -1), not at the zero offset
    # worst-case delta (765 << 10) x 64x64 window exceeds 2^32: the reference sums wrap, so must we
    a = np.zeros(128 * 64 * 3 // 2, np.uint8)
    b = np.full_like(a, 255)
    g2 = oracle.make_geom(0, 64, 128)
    sums, _ = oracle.calc_delta_sums(a, b, np.zeros((2, 64, 128), np.int16), g2, 64, 5, 0, 0, delta_scalar=10, neighbor_scalar=0)
    assert int(sums[2, 0, 0]) == (765 * 1024 * 64 * 64) % (1 << 32)
