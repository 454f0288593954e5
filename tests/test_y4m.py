"""YUV4MPEG2 reader / writer (host-side re-layout only, no device).  SURVEY.md section 8(f) row 4."""
import io

import numpy as np
import pytest

from hopperrender_amd import y4m


def _planes(H, W, bits, seed):
    rng = np.random.default_rng(seed)
    dt = np.uint16 if bits > 8 else np.uint8
    hi = 1 << bits
    return (rng.integers(0, hi, (H, W)).astype(dt), rng.integers(0, hi, (H // 2, W // 2)).astype(dt),
            rng.integers(0, hi, (H // 2, W // 2)).astype(dt))


@pytest.mark.parametrize("bits", [8, 10])
def test_layout_round_trip(bits):
    H, W = 6, 8
    y, u, v = _planes(H, W, bits, 1)
    hdr = bits > 8
    f = y4m.planar_to_semiplanar(y, u, v, hdr)
    assert f.dtype == (np.uint16 if hdr else np.uint8) and f.size == H * W * 3 // 2
    sh = 6 if hdr else 0
    assert (f[:H * W].reshape(H, W) == (y.astype(np.uint32) << sh)).all()
    uv = f[H * W:].reshape(H // 2, W)
    assert (uv[:, 0::2] == (u.astype(np.uint32) << sh)).all() and (uv[:, 1::2] == (v.astype(np.uint32) << sh)).all()
    y2, u2, v2 = y4m.semiplanar_to_planar(f, H, W, hdr)
    assert (y2 == y).all() and (u2 == u).all() and (v2 == v).all()


@pytest.mark.parametrize("cs,bits", [("420jpeg", 8), ("420mpeg2", 8), ("420", 8), ("420p10", 10)])
def test_reader_parses_stream(cs, bits):
    H, W, n = 4, 6, 3
    frames = [_planes(H, W, bits, k) for k in range(n)]
    dt = "<u2" if bits > 8 else "u1"
    blob = f"YUV4MPEG2 W{W} H{H} F30000:1001 Ip A1:1 C{cs} XCOMMENT\n".encode()
    for y, u, v in frames:
        blob += b"FRAME\n" + y.astype(dt).tobytes() + u.astype(dt).tobytes() + v.astype(dt).tobytes()
    blob += b"FRAME\n" + b"\0" * 5      # trailing partial frame is dropped like the raw reader does
    r = y4m.Y4MReader(io.BytesIO(blob))
    assert (r.width, r.height, r.fps_num, r.fps_den, r.hdr) == (W, H, 30000, 1001, bits > 8)
    assert abs(r.fps - 29.97) < 0.001 and "Ip" in r.extra and "A1:1" in r.extra
    got = list(r)
    assert len(got) == n
    for g, (y, u, v) in zip(got, frames):
        assert (g == y4m.planar_to_semiplanar(y, u, v, bits > 8)).all()


@pytest.mark.parametrize("hdr", [False, True])
def test_writer_reader_round_trip(hdr):
    H, W = 8, 12
    rng = np.random.default_rng(7)
    frames = [y4m.planar_to_semiplanar(*_planes(H, W, 10 if hdr else 8, k), hdr) for k in range(4)]
    bio = io.BytesIO()
    w = y4m.Y4MWriter(bio, W, H, 60, 1, hdr, extra=("Ip",))
    for f in frames:
        w.write(f)
    bio.seek(0)
    head = bio.getvalue().split(b"\n", 1)[0]
    assert head.startswith(b"YUV4MPEG2 W12 H8 F60:1 Ip C420")
    r = y4m.Y4MReader(bio)
    got = list(r)
    assert len(got) == 4 and all((a == b).all() for a, b in zip(got, frames))


@pytest.mark.parametrize("head", [b"RIFF W4 H4\n", b"YUV4MPEG2 W4 H4 C444\n", b"YUV4MPEG2 W5 H4 C420\n", b"YUV4MPEG2 H4 C420\n"])
def test_reader_rejects(head):
    with pytest.raises(y4m.Y4MError):
        y4m.Y4MReader(io.BytesIO(head))


def test_missing_frame_marker():
    r = y4m.Y4MReader(io.BytesIO(b"YUV4MPEG2 W4 H4 C420\nFRAMX\n" + b"\0" * 24))
    with pytest.raises(y4m.Y4MError):
        next(r)


def test_random_access_clip_and_worker_file_layout(tmp_path):
    """The multi-GPU CLI's workers read source frames by index (cli._Clip) and write output frames at their final offsets:
    raw = i x frame bytes; .y4m = header + i x (6 + frame bytes), the header being exactly what Y4MWriter emits."""
    import io
    from hopperrender_amd import cli, synth
    from hopperrender_amd.y4m import Y4MWriter
    H, W = 36, 64
    for hdr in (False, True):
        sc = synth.Scene(H, W, hdr, 3)
        frames = [sc.frame(k) for k in range(5)]
        if hdr:
            frames = [(f >> 6) << 6 for f in frames]          # .y4m keeps the 10-bit code only
        raw, y4m = tmp_path / f"c{int(hdr)}.bin", tmp_path / f"c{int(hdr)}.y4m"
        with open(raw, "wb") as f:
            for x in frames:
                f.write(x.tobytes())
        with open(y4m, "wb") as f:
            w = Y4MWriter(f, W, H, 24000, 1001, hdr)
            for x in frames:
                w.write(x)
        for path, kw in ((raw, dict(width=W, height=H, hdr=hdr, source_fps=None)), (y4m, dict(width=None, height=None, hdr=False, source_fps=None))):
            clip = cli._Clip(str(path), **kw)
            assert (clip.n_frames, clip.width, clip.height, clip.hdr) == (5, W, H, hdr)
            buf = np.empty(clip.n_el, dtype=clip.dt)
            for k in (3, 0, 4):
                clip.read_into(k, buf)
                assert (buf == frames[k]).all(), (path, k)
        clip = cli._Clip(str(y4m), None, None, False, None)
        ref = io.BytesIO()
        Y4MWriter(ref, W, H, 60, 1, hdr, clip.extra)
        assert cli._y4m_header(clip, 60.0) == ref.getvalue()
        assert len(open(y4m, "rb").read()) == clip.data0 + 5 * (6 + clip.frame_bytes)
