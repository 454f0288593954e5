"""Seeded synthetic NV12 / P010 frame sequences (SURVEY.md section 8(d) "Synthetic inputs").

Content: band-limited noise background (uniform noise box-filtered 9x9) that translates by a
global (+7, -3) px per frame, 12 opaque rectangles with individual velocities up to +-48 px per
frame, +-1 LSB temporal noise.  SDR = NV12 8-bit full range; HDR = P010 (10-bit code in the top
bits, low 6 bits zero, codes 64..940).  numpy only; used by tests, bench.py and the golden-vector
script -- inputs are regenerated from the seed, never stored.
"""
import numpy as np

GLOBAL_MOTION = (7, -3)  # (dx, dy) px per frame


def _box9(a):
    """9x9 box filter with wrap-around borders (wrap keeps the global translation exact)."""
    k = 9
    pad = k // 2
    ap = np.pad(a, ((pad, pad), (pad, pad)), mode="wrap").astype(np.float32)
    c = np.cumsum(ap, axis=0)
    c = np.concatenate([np.zeros((1, c.shape[1]), np.float32), c], axis=0)
    r = c[k:, :] - c[:-k, :]
    c = np.cumsum(r, axis=1)
    c = np.concatenate([np.zeros((c.shape[0], 1), np.float32), c], axis=1)
    r = c[:, k:] - c[:, :-k]
    return r / float(k * k)


class Scene:
    """A deterministic moving scene; frame(k) renders time step k."""

    def __init__(self, H, W, hdr=False, seed=1234, n_rects=12, max_rect_speed=48, in_stride=0, global_motion=GLOBAL_MOTION):
        assert H % 2 == 0 and W % 2 == 0
        self.H, self.W, self.hdr = H, W, bool(hdr)
        self.global_motion = global_motion
        self.stride = in_stride if in_stride > 0 else W
        self.seed = seed
        rng = np.random.default_rng(seed)
        # three 8-bit background planes at full resolution (Y, U, V), contrast-stretched
        self.bg = []
        for _ in range(3):
            n = _box9(rng.integers(0, 256, size=(H, W)).astype(np.float32))
            n = (n - n.mean()) * 6.0 + 128.0
            self.bg.append(np.clip(n, 0, 255).astype(np.uint8))
        self.rects = []
        for _ in range(n_rects):
            rw = int(rng.integers(max(W // 24, 4), max(W // 5, 8)))
            rh = int(rng.integers(max(H // 24, 4), max(H // 5, 8)))
            x0 = int(rng.integers(0, W - rw))
            y0 = int(rng.integers(0, H - rh))
            vx = int(rng.integers(-max_rect_speed, max_rect_speed + 1))
            vy = int(rng.integers(-max_rect_speed // 2, max_rect_speed // 2 + 1))
            col = rng.integers(16, 240, size=3)
            self.rects.append((x0, y0, rw, rh, vx, vy, tuple(int(c) for c in col)))

    def _planes8(self, k):
        H, W = self.H, self.W
        dx, dy = self.global_motion[0] * k, self.global_motion[1] * k
        planes = [np.roll(p, (dy, dx), axis=(0, 1)).copy() for p in self.bg]
        for (x0, y0, rw, rh, vx, vy, col) in self.rects:
            x = (x0 + vx * k) % W
            y = (y0 + vy * k) % H
            x1, y1 = min(x + rw, W), min(y + rh, H)
            for p, c in zip(planes, col):
                p[y:y1, x:x1] = c
        rng = np.random.default_rng(self.seed * 1000003 + k)
        out = []
        for p in planes:
            n = rng.integers(-1, 2, size=p.shape)
            out.append(np.clip(p.astype(np.int16) + n, 0, 255).astype(np.uint8))
        return out

    def frame(self, k):
        """Contiguous NV12 (uint8) or P010 (uint16) frame k: Y plane [H][stride] then UV [H/2][stride]."""
        return self._pack(*self._planes8(k))

    def _pack(self, y8, u8, v8):
        H, W, S = self.H, self.W, self.stride
        u8, v8 = u8[::2, ::2], v8[::2, ::2]
        if self.hdr:
            def to10(p):  # 8-bit -> 10-bit code 64..940, stored in the top bits of 16
                c = 64 + (p.astype(np.uint32) * 876 + 127) // 255
                return (c << 6).astype(np.uint16)
            y, u, v = to10(y8), to10(u8), to10(v8)
            dt = np.uint16
        else:
            y, u, v = y8, u8, v8
            dt = np.uint8
        f = np.zeros((H + H // 2, S), dtype=dt)
        f[:H, :W] = y
        f[H:, 0:W:2] = u
        f[H:, 1:W:2] = v
        return f.reshape(-1)


def frame_pair(H, W, hdr=False, seed=1234, in_stride=0):
    s = Scene(H, W, hdr, seed, in_stride=in_stride)
    return s.frame(0), s.frame(1)


def random_frame(H, W, hdr=False, seed=0, in_stride=0):
    """White-noise frame (every code value), for copy/levels and index-path tests."""
    S = in_stride if in_stride > 0 else W
    rng = np.random.default_rng(seed)
    if hdr:
        return rng.integers(0, 65536, size=(H + H // 2) * S, dtype=np.uint16)
    return rng.integers(0, 256, size=(H + H // 2) * S, dtype=np.uint8)


# ------------------------------------------------------------------------------------------------
# Content classes of SURVEY.md 8(d) "extra cases" (bench.py --scene, tests): the reference's cost does not depend on the pixels
# (opticalFlowCalcSDR.cpp:44-139 issues the same grids whatever they are); this build's does -- the staged warp's window fit and the
# chain's SAD reuse follow the motion -- so every headline number names the content it ran on.
# ------------------------------------------------------------------------------------------------
SCENES = ("bench", "static", "pan64", "chaotic", "cut")


class ContentScene:
    """frame(k) of one content class:
         bench    the default Scene: global (+7, -3) px per frame + 12 rectangles up to +-48 px per frame
         static   one frame repeated (flow must be zero, every window reuses)
         pan64    pure translation by 64 px per frame, no rectangles
         chaotic  full-range noise (every code value, 3 x 3 box filtered so that matches exist) whose 16 x 16 blocks each move by their own random
                  displacement up to +-96 px per frame: no two neighbouring windows agree, the hostile case for window fit and SAD reuse
         cut      every consecutive pair of frames is a hard scene cut (two alternating scenes)"""

    def __init__(self, name, H, W, hdr=False, seed=1234, in_stride=0):
        assert name in SCENES, name
        self.name, self.H, self.W, self.hdr, self.seed = name, H, W, bool(hdr), seed
        if name == "pan64":
            self.a = Scene(H, W, hdr, seed, n_rects=0, in_stride=in_stride, global_motion=(64, 0))
        else:
            self.a = Scene(H, W, hdr, seed, in_stride=in_stride)
        self.b = Scene(H, W, hdr, seed + 999, in_stride=in_stride) if name == "cut" else None
        self._chaos = {}

    def _chaotic_planes(self, k):
        if k in self._chaos:
            return self._chaos[k]
        H, W = self.H, self.W
        if k == 0:
            rng = np.random.default_rng(self.seed * 7919 + 1)
            planes = []
            for _ in range(3):
                n = rng.integers(0, 256, size=(H, W)).astype(np.float32)
                ap = np.pad(n, 1, mode="wrap")
                n = sum(ap[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)) / 9.0
                n = (n - 128.0) * 3.0 + 128.0
                planes.append(np.clip(n, 0, 255).astype(np.uint8))
        else:
            prev = self._chaotic_planes(k - 1)
            rng = np.random.default_rng(self.seed * 7919 + 1 + k)
            by, bx = (H + 15) // 16, (W + 15) // 16
            dx = np.repeat(np.repeat(rng.integers(-96, 97, size=(by, bx)), 16, axis=0), 16, axis=1)[:H, :W]
            dy = np.repeat(np.repeat(rng.integers(-96, 97, size=(by, bx)), 16, axis=0), 16, axis=1)[:H, :W]
            yy, xx = np.mgrid[0:H, 0:W]
            sy, sx = (yy - dy) % H, (xx - dx) % W
            planes = [p[sy, sx] for p in prev]
        self._chaos = {k: planes} if k == 0 else {k - 1: self._chaos.get(k - 1, prev), k: planes}
        return planes

    def frame(self, k):
        if self.name == "static":
            return self.a.frame(0)
        if self.name == "cut":
            return (self.a if k % 2 == 0 else self.b).frame(k)
        if self.name == "chaotic":
            return self.a._pack(*self._chaotic_planes(k))
        return self.a.frame(k)


def scene_pair(name, H, W, hdr=False, seed=1234):
    """Frames N-1, N of one pair of content class `name` (frames 1 and 2 of the sequence, as tools/chain_time.py runs them)."""
    s = ContentScene(name, H, W, hdr, seed)
    return s.frame(1), s.frame(2)
