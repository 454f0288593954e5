"""YUV4MPEG2 (.y4m) reader / writer for the CLI (on-disk formats either side of the
path).  The filter's own pin formats are NV12 (8-bit) and P010 (10-bit code in the top bits of 16,
reference HopperRender.cpp:603-625 media types); Y4M stores planar 4:2:0, so frames are re-laid here:

    C420 / C420jpeg / C420mpeg2 / C420paldv   Y, U, V planes of uint8       <->  NV12
    C420p10                                   Y, U, V planes of LE uint16   <->  P010 (code << 6)

Only the host-side byte shuffling lives here; nothing in this module touches the device.
"""
import numpy as np

_CS8 = ("420", "420jpeg", "420mpeg2", "420paldv")


class Y4MError(ValueError):
    pass


def planar_to_semiplanar(y, u, v, hdr):
    """Y[H][W], U[H/2][W/2], V[H/2][W/2] -> contiguous NV12 / P010 frame (1-D, Y plane then interleaved UV)."""
    H, W = y.shape
    dt = np.uint16 if hdr else np.uint8
    out = np.empty(H * W * 3 // 2, dtype=dt)
    out[:H * W] = y.reshape(-1)
    uv = out[H * W:].reshape(H // 2, W // 2, 2)
    uv[:, :, 0] = u
    uv[:, :, 1] = v
    if hdr:
        out <<= 6    # 10-bit code -> top bits, low 6 bits zero (P010)
    return out


def semiplanar_to_planar(frame, H, W, hdr):
    """Inverse of planar_to_semiplanar; P010 is cut back to its 10-bit code (>> 6)."""
    f = np.asarray(frame).reshape(-1)
    if hdr:
        f = f >> 6
    y = f[:H * W].reshape(H, W)
    uv = f[H * W:H * W * 3 // 2].reshape(H // 2, W // 2, 2)
    return y, np.ascontiguousarray(uv[:, :, 0]), np.ascontiguousarray(uv[:, :, 1])


class Y4MReader:
    """Iterates the frames of a .y4m stream as NV12 / P010 arrays."""

    def __init__(self, fileobj):
        self.f = fileobj
        head = self._line()
        tok = head.split(b" ")
        if tok[0] != b"YUV4MPEG2":
            raise Y4MError("not a YUV4MPEG2 stream")
        self.width = self.height = 0
        self.fps_num, self.fps_den = 24000, 1001
        self.colourspace = "420"
        self.extra = []           # interlacing / aspect / comment tokens carried over to the writer
        for t in tok[1:]:
            if not t:
                continue
            k, val = t[:1], t[1:].decode("ascii")
            if k == b"W":
                self.width = int(val)
            elif k == b"H":
                self.height = int(val)
            elif k == b"F":
                n, d = val.split(":")
                self.fps_num, self.fps_den = int(n), int(d)
            elif k == b"C":
                self.colourspace = val
            else:
                self.extra.append(t.decode("ascii"))
        if self.colourspace in _CS8:
            self.hdr = False
        elif self.colourspace == "420p10":
            self.hdr = True
        else:
            raise Y4MError(f"unsupported colourspace C{self.colourspace} (need 4:2:0, 8 or 10 bit)")
        if self.width <= 0 or self.height <= 0 or self.width % 2 or self.height % 2:
            raise Y4MError(f"bad frame size {self.width}x{self.height}")

    @property
    def fps(self):
        return self.fps_num / self.fps_den

    def _line(self):
        b = bytearray()
        while True:
            c = self.f.read(1)
            if not c:
                if b:
                    raise Y4MError("truncated header")
                return b""
            if c == b"\n":
                return bytes(b)
            b += c
            if len(b) > 4096:
                raise Y4MError("header line too long")

    def __iter__(self):
        return self

    def __next__(self):
        head = self._line()
        if not head:
            raise StopIteration
        if not head.startswith(b"FRAME"):
            raise Y4MError("missing FRAME marker")
        H, W = self.height, self.width
        dt = np.dtype("<u2") if self.hdr else np.dtype(np.uint8)
        n = H * W * 3 // 2
        buf = self.f.read(n * dt.itemsize)
        if len(buf) < n * dt.itemsize:
            raise StopIteration      # trailing partial frame: same policy as the raw reader
        p = np.frombuffer(buf, dtype=dt)
        y = p[:H * W].reshape(H, W)
        u = p[H * W:H * W * 5 // 4].reshape(H // 2, W // 2)
        v = p[H * W * 5 // 4:].reshape(H // 2, W // 2)
        return planar_to_semiplanar(y, u, v, self.hdr)


class Y4MWriter:
    """Writes NV12 / P010 frames as a .y4m stream."""

    def __init__(self, fileobj, width, height, fps_num, fps_den, hdr, extra=()):
        self.f, self.width, self.height, self.hdr = fileobj, width, height, hdr
        cs = "420p10 XYSCSS=420P10" if hdr else "420jpeg"
        toks = [f"W{width}", f"H{height}", f"F{fps_num}:{fps_den}"] + [e for e in extra if not e.startswith("XYSCSS")] + [f"C{cs}"]
        self.f.write(("YUV4MPEG2 " + " ".join(toks) + "\n").encode("ascii"))

    def write(self, frame):
        y, u, v = semiplanar_to_planar(frame, self.height, self.width, self.hdr)
        dt = np.dtype("<u2") if self.hdr else np.dtype(np.uint8)
        self.f.write(b"FRAME\n")
        for p in (y, u, v):
            self.f.write(np.ascontiguousarray(p, dtype=dt).tobytes())
