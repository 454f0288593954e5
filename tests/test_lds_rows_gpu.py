"""GPU: the Y steps that read their candidate rows from LDS (csrc/hf_flow.hip ysads_tile_lds / ysads_win8_lds: windows >= 8, full
tiles, R = 16, rs <= 4, candidate rows that need no reflection -- everything else keeps the per-candidate gathers) against the CPU
oracle, on inputs chosen for the decisions that path makes per workgroup / per wave:
  * white noise: offsets all over the +-64 range, so tiles near the top and bottom edge mix staged and reflected windows;
  * a frame shifted vertically by a known amount: every window carries a large Y offset, rows far from the tile;
  * grids that are not multiples of the tiles (partial tiles take the gathers); every resolution scalar 0 .. 4 (256-line to 8K frames);
  * a batch of 6 (the block-per-lane mapping of the fine levels) next to the single context.
Bar: bit-exact offsets, blurred flow and total frame delta (calcDeltaSumsKernelSDR.h:61-190, determineLowestLayerKernelSDR.h:16-26)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shifted(frame, H, S, dy):
    """The frame moved down by dy luma rows (dy even; rows wrap around)."""
    y = frame[:H * S].reshape(H, S)
    uv = frame[H * S:].reshape(H // 2, S)
    return np.concatenate([np.roll(y, dy, axis=0).ravel(), np.roll(uv, dy // 2, axis=0).ravel()])


def _frames(kind, H, W, hdr, seed):
    from hopperrender_amd import synth
    if kind == "noise":
        return [synth.random_frame(H, W, hdr, seed=seed + i) for i in range(3)]
    if kind.startswith("shift"):
        dy = int(kind[5:])
        sc = synth.Scene(H, W, hdr, seed=seed)
        a = sc.frame(0)
        return [a, a, _shifted(a, H, W, dy)]
    sc = synth.Scene(H, W, hdr, seed=seed, max_rect_speed=48)
    return [sc.frame(i) for i in range(3)]


CASES = [
    # (hdr, H, W, max_calc_res, content)                grid          rs
    (0, 1080, 1920, 270, "noise"),                    # 480 x 270     2   BASELINE config 2's geometry
    (1, 2160, 3840, 270, "noise"),                    # 480 x 270     3   the bench's geometry
    (1, 2160, 3840, 270, "shift40"),
    (0, 1080, 1920, 270, "shift-56"),
    (0, 544, 960, 136, "scene"),                      # 240 x 136     2   partial tiles right (240 = 7.5 x 32) and bottom
    (1, 1088, 1920, 136, "noise"),                    # 240 x 136     3
    (0, 1080, 1920, 180, "scene"),                    # 240 x 135     3   135 grid rows: the last tile row has 7
    (1, 1200, 2080, 300, "shift24"),                  # 520 x 300     2   wider than 512: four levels of large windows
    (0, 256, 480, 270, "noise"),                      # 480 x 256     0   the grid is the frame: one phase, one residue class
    (1, 540, 960, 270, "shift-20"),                   # 480 x 270     1
    (0, 540, 960, 270, "noise"),                      # 480 x 270     1
    (0, 4320, 7680, 270, "noise"),                    # 480 x 270     4   8K: seven residue classes
]


@pytest.mark.parametrize("hdr,H,W,max_res,content", CASES)
def test_flow_with_lds_rows_matches_oracle(native_lib, hdr, H, W, max_res, content):
    from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR
    from oracle import oracle
    f = _frames(content, H, W, bool(hdr), seed=4100 + H + len(content))
    g = oracle.make_geom(hdr, H, W, 0, 0, max_res)
    cls = OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR
    c = cls(H, W, 0, 0, 8, 6, 0.0, 255.0, max_res, search_radius=16)
    for x in f:
        c.updateFrame(x)
    c.calculateOpticalFlow()
    c.sync()
    off, blur, tot, _ = oracle.calculate_optical_flow(f[1], f[2], g, 16, 0, 8, 6, 4)
    assert (c.readOffsets() == off).all(), (hdr, H, W, content, int((c.readOffsets() != off).sum()))
    assert (c.readBlurredFlow(1) == blur).all()
    assert c.m_totalFrameDelta == tot
    if content.startswith("shift"):   # the test means what it says: most windows found the vertical shift
        dy = int(content[5:])
        assert (np.abs(off[1].astype(int) + dy) <= 2).mean() > 0.5, np.unique(off[1], return_counts=True)
    c.close()


def test_batch_of_six_with_lds_rows_matches_oracle(native_lib):
    """Six pairs in one set of launches (more than four: the fine levels use a block per lane): every member against the oracle."""
    from hopperrender_amd import capi
    from hopperrender_amd.calc import FlowBatch, OpticalFlowCalcSDR
    from oracle import oracle
    H, W = 1080, 1920
    g = oracle.make_geom(0, H, W, 0, 0, 270)
    kinds = ["noise", "shift32", "scene", "shift-48", "noise", "scene"]
    cs, fs = [], []
    for i, kind in enumerate(kinds):
        f = _frames(kind, H, W, False, seed=5200 + 17 * i)
        c = OpticalFlowCalcSDR(H, W, 0, 0, 8, 6, 0.0, 255.0, 270, search_radius=16, flags=capi.HF_FLAG_ASYNC)
        for x in f:
            c.updateFrame(x)
        cs.append(c); fs.append(f)
    b = FlowBatch(cs)
    b.calculateOpticalFlow()
    cs[0].sync()
    for c, f in zip(cs, fs):
        off, blur, tot, _ = oracle.calculate_optical_flow(f[1], f[2], g, 16, 0, 8, 6, 4)
        assert (c.readOffsets() == off).all() and (c.readBlurredFlow(1) == blur).all() and c.m_totalFrameDelta == tot
    b.close()
    for c in cs:
        c.close()
