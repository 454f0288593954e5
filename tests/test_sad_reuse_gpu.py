"""GPU: the chain's exact cross-step SAD reuse (csrc/hf_flow.hip "SAD TABLES": a window whose parent chose d = 0 at the two preceding
steps sums the per-block candidate SADs the last computing step left instead of gathering the phase plane again) against the CPU oracle
and against the same library with HF_FLAG_NO_SAD_REUSE, on inputs where the decision flips window by window:
  * the bench's content classes (synth.ContentScene: bench, static = every window reuses, pan64, chaotic / cut = hardly any does);
  * noise patches inside a static frame: reuse and recomputation alternate between neighbouring windows at every level;
  * grids with partial tiles (their windows never touch the tables), a 240-wide grid whose last 16 columns are a full 16-wide tile of the
    row-per-lane level-2 mapping inside a partial 32 x 32 tile, grids whose first small level is 16 (no level before it to reuse);
  * R = 5 (no full tiles at all), other delta / neighbour scalars, fewer iterations;
  * batches of 2 (a row per lane at the two finest levels) and of 6 (a block per lane).
Bar: bit-exact offsets, blurred flow and total frame delta (calcDeltaSumsKernelSDR.h:61-190, determineLowestLayerKernelSDR.h:16-26,
adjustOffsetArrayKernelSDR.h:11-19, opticalFlowCalcSDR.cpp:68-111)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _patched(frame_a, H, S, hdr, seed, n=40):
    """frame_a with n rectangular patches (8 .. 96 px) of fresh noise: the rest of the frame is static."""
    rng = np.random.default_rng(seed)
    f = frame_a.copy()
    y = f[:H * S].reshape(H, S)
    uv = f[H * S:].reshape(H // 2, S)
    hi = 65536 if hdr else 256
    for _ in range(n):
        ph, pw = int(rng.integers(4, min(49, H // 4))) * 2, int(rng.integers(4, min(49, S // 4))) * 2
        y0, x0 = int(rng.integers(0, (H - ph) // 2)) * 2, int(rng.integers(0, (S - pw) // 2)) * 2
        y[y0:y0 + ph, x0:x0 + pw] = rng.integers(0, hi, size=(ph, pw))
        uv[y0 // 2:(y0 + ph) // 2, x0:x0 + pw] = rng.integers(0, hi, size=(ph // 2, pw))
    return f


def _frames(kind, H, W, hdr, seed):
    from hopperrender_amd import synth
    if kind == "patches":
        a = synth.Scene(H, W, hdr, seed=seed).frame(0)
        return [a, a, _patched(a, H, W, hdr, seed + 1)]
    if kind == "noise":
        return [synth.random_frame(H, W, hdr, seed=seed + i) for i in range(3)]
    sc = synth.ContentScene(kind, H, W, hdr, seed)
    return [sc.frame(i) for i in range(3)]


def _run(cls, H, W, max_res, f, R=16, delta=8, nb=6, iterations=0, flags=None):
    from hopperrender_amd import capi
    flags = capi.HF_FLAG_SAD_REUSE_ALWAYS if flags is None else flags      # (the default would drop the tables on hostile content after a chain or two)
    c = cls(H, W, 0, 0, delta, nb, 0.0, 255.0, max_res, search_radius=R, iterations=iterations, flags=flags)
    for x in f:
        c.updateFrame(x)
    c.calculateOpticalFlow()
    c.sync()
    r = (c.readOffsets(), c.readBlurredFlow(1), c.m_totalFrameDelta)
    c.close()
    return r


CASES = [
    # (hdr, H, W, max_calc_res, content, R, delta, nb, iterations)
    (0, 1080, 1920, 270, "bench", 16, 8, 6, 0),
    (1, 2160, 3840, 270, "bench", 16, 8, 6, 0),
    (0, 1080, 1920, 270, "static", 16, 8, 6, 0),
    (0, 1080, 1920, 270, "pan64", 16, 8, 6, 0),
    (0, 1080, 1920, 270, "chaotic", 16, 8, 6, 0),
    (1, 2160, 3840, 270, "chaotic", 16, 8, 6, 0),
    (0, 1080, 1920, 270, "cut", 16, 8, 6, 0),
    (0, 1080, 1920, 270, "patches", 16, 8, 6, 0),
    (1, 2160, 3840, 270, "patches", 16, 8, 10, 0),
    (0, 1080, 1920, 270, "patches", 16, 3, 0, 0),
    (0, 1080, 1920, 270, "patches", 5, 8, 6, 0),         # R = 5: the reference's starting radius, no full tiles
    (0, 1080, 1920, 270, "bench", 11, 8, 6, 0),
    (0, 544, 960, 136, "patches", 16, 8, 6, 0),          # 240 x 136 grid: partial tiles right and bottom; 240 = 7.5 x 32
    (1, 1088, 1920, 136, "bench", 16, 8, 6, 0),
    (0, 1080, 1920, 180, "patches", 16, 8, 6, 0),        # 240 x 135
    (0, 64, 64, 32, "patches", 16, 8, 6, 0),             # 32 x 32 grid: the chain's first level is 16
    (0, 128, 128, 64, "bench", 16, 8, 6, 0),             # 64 x 64 grid: first level 32
    (0, 256, 480, 270, "patches", 16, 8, 6, 0),          # rs = 0
    (1, 540, 960, 270, "patches", 16, 8, 6, 0),          # rs = 1
    (0, 1080, 1920, 270, "patches", 16, 8, 6, 6),        # six levels: the chain ends at windows of 8
    (0, 1080, 1920, 270, "bench", 16, 8, 6, 4),          # ends at 32: tables written, never read
]


@pytest.mark.parametrize("hdr,H,W,max_res,content,R,delta,nb,iterations", CASES)
def test_reuse_matches_oracle_and_recompute(native_lib, hdr, H, W, max_res, content, R, delta, nb, iterations):
    from hopperrender_amd import capi
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    f = _frames(content, H, W, bool(hdr), seed=7100 + H + len(content))
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    a = _run(cls, H, W, max_res, f, R, delta, nb, iterations)
    b = _run(cls, H, W, max_res, f, R, delta, nb, iterations, flags=capi.HF_FLAG_NO_SAD_REUSE)
    g = oracle.make_geom(hdr, H, W, 0, 0, max_res)
    off, blur, tot, oob = oracle.calculate_optical_flow(f[1], f[2], g, R, iterations, delta, nb, 4)
    tag = (hdr, H, W, content, R)
    assert (b[0] == off).all() and (b[1] == blur).all() and b[2] == tot, ("recompute", tag, int((b[0] != off).sum()))
    assert (a[0] == off).all(), ("reuse", tag, int((a[0] != off).sum()))
    assert (a[1] == blur).all() and a[2] == tot, ("reuse", tag)


@pytest.mark.parametrize("n", [2, 6])
def test_batches_with_reuse_match_oracle(native_lib, n):
    """Batches of 2 (a row per lane at levels 4 / 2) and 6 (a block per lane), members of different content; the graph replay twice."""
    from hopperrender_amd import capi
    from hopperrender_amd.calc import FlowBatch, OpticalFlowCalcSDR
    from oracle import oracle
    H, W = 1080, 1920
    g = oracle.make_geom(0, H, W, 0, 0, 270)
    kinds = ["patches", "bench", "chaotic", "static", "pan64", "cut"][:n]
    cs, fs = [], []
    for i, kind in enumerate(kinds):
        f = _frames(kind, H, W, False, seed=8200 + 31 * i)
        c = OpticalFlowCalcSDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_ASYNC | capi.HF_FLAG_SAD_REUSE_ALWAYS)
        for x in f:
            c.updateFrame(x)
        cs.append(c); fs.append(f)
    b = FlowBatch(cs)
    for _ in range(2):
        b.calculateOpticalFlow()
        cs[0].sync()
        for c, f in zip(cs, fs):
            off, blur, tot, _ = oracle.calculate_optical_flow(f[1], f[2], g, 16, 0, 8, 6, 4)
            assert (c.readOffsets() == off).all() and (c.readBlurredFlow(1) == blur).all() and c.m_totalFrameDelta == tot
    b.close()
    for c in cs:
        c.close()
