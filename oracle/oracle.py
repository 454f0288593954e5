"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/libhf_oracle.so (the plain-C CPU restatement, hf_oracle.c) and a
helper that drives oracle/_ref/ref_runner (the reference itself, needs an OpenCL GPU).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhf_oracle.so")
REF_RUNNER = os.path.join(HERE, "_ref", "ref_runner")


class Geom(C.Structure):
    _fields_ = [("hdr", C.c_int), ("H", C.c_int), ("W", C.c_int), ("in_stride", C.c_int),
                ("out_stride", C.c_int), ("rs", C.c_int), ("lw", C.c_int), ("lh", C.c_int)]


class Diag(C.Structure):
    _fields_ = [("oob_samples", C.c_uint64)]


def build(force=False):
    """Compile the C restatement (and, when /root/reference is present, oracle/_ref)."""
    src = [os.path.join(HERE, "hf_oracle.c"), os.path.join(HERE, "hf_oracle.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/HopperRender"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.hfo_make_geom.argtypes = [C.POINTER(Geom), i, i, i, i, i, i]
        L.hfo_initial_window.argtypes = [i, i]; L.hfo_initial_window.restype = i
        L.hfo_iterations.argtypes = [i, i]; L.hfo_iterations.restype = i
        L.hfo_rel_offset.argtypes = [i, i]; L.hfo_rel_offset.restype = i
        L.hfo_calc_delta_sums.argtypes = [vp, vp, vp, vp, C.POINTER(Geom), i, i, i, i, i, i, C.POINTER(Diag)]
        L.hfo_determine_lowest_layer.argtypes = [vp, vp, i, i, i, i]
        L.hfo_adjust_offsets.argtypes = [vp, vp, i, i, i, i, i]
        L.hfo_blur_flow.argtypes = [vp, vp, i, i, i]
        L.hfo_calculate_optical_flow.argtypes = [vp, vp, C.POINTER(Geom), i, i, i, i, i, vp, vp,
                                                 C.POINTER(C.c_uint32), C.POINTER(Diag)]
        L.hfo_warp_frames.argtypes = [vp, vp, vp, vp, C.POINTER(Geom), f, i, f, f]
        L.hfo_copy_frame.argtypes = [vp, vp, C.POINTER(Geom), f, f]
        L.hfo_set_blend_flavour.argtypes = [i]
        L.hfo_set_levels_flavour.argtypes = [i]
        L.hfo_set_rcp_override.argtypes = [i, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_flavour(blend=1, levels=1, rcp=None):
    """Select the fp32 flavour of blend/levels (hf_oracle.h); rcp = {y: device_rcp(y)} overrides."""
    lib().hfo_set_blend_flavour(blend)
    lib().hfo_set_levels_flavour(levels)
    if rcp:
        ys = np.array(list(rcp.keys()), dtype=np.float32)
        rs = np.array(list(rcp.values()), dtype=np.float32)
        lib().hfo_set_rcp_override(len(ys), _p(ys), _p(rs))
    else:
        lib().hfo_set_rcp_override(0, None, None)


def make_geom(hdr, H, W, in_stride=0, out_stride=0, max_calc_res=270):
    g = Geom()
    lib().hfo_make_geom(C.byref(g), int(hdr), H, W, in_stride, out_stride, max_calc_res)
    return g


def frame_dtype(hdr):
    return np.uint16 if hdr else np.uint8


def in_elems(g):
    return g.H * g.in_stride + (g.H // 2) * g.in_stride


def out_elems(g):
    return g.H * g.out_stride + (g.H // 2) * g.out_stride


def calc_delta_sums(f1, f2, offsets, g, window, R, iteration, step, delta_scalar=8, neighbor_scalar=6):
    N = g.lw * g.lh
    sums = np.zeros(R * N, dtype=np.uint32)
    d = Diag()
    lib().hfo_calc_delta_sums(_p(sums), _p(f1), _p(f2), _p(offsets), C.byref(g), window, R, iteration, step,
                              delta_scalar, neighbor_scalar, C.byref(d))
    return sums.reshape(R, g.lh, g.lw), d.oob_samples


def blur_flow(offsets, g, radius=4):
    out = np.empty_like(offsets)
    lib().hfo_blur_flow(_p(offsets), _p(out), g.lh, g.lw, radius)
    return out


def calculate_optical_flow(f1, f2, g, R, iterations=0, delta_scalar=8, neighbor_scalar=6, blur_radius=4):
    """Returns (offsets[2,lh,lw], blurred[2,lh,lw], total_frame_delta, oob_samples)."""
    off = np.zeros((2, g.lh, g.lw), dtype=np.int16)
    blur = np.zeros((2, g.lh, g.lw), dtype=np.int16)
    tot = C.c_uint32(0)
    d = Diag()
    lib().hfo_calculate_optical_flow(_p(f1), _p(f2), C.byref(g), R, iterations, delta_scalar, neighbor_scalar,
                                     blur_radius, _p(off), _p(blur), C.byref(tot), C.byref(d))
    return off, blur, tot.value, d.oob_samples


def warp_frames(f12, f21, flow, g, t, mode=2, black=0.0, white=255.0, out=None):
    if out is None:
        out = np.zeros(out_elems(g), dtype=frame_dtype(g.hdr))
    lib().hfo_warp_frames(_p(f12), _p(f21), _p(flow), _p(out), C.byref(g), float(t), mode, float(black), float(white))
    return out


def copy_frame(src, g, black=0.0, white=255.0, out=None):
    if out is None:
        out = np.zeros(out_elems(g), dtype=frame_dtype(g.hdr))
    lib().hfo_copy_frame(_p(src), _p(out), C.byref(g), float(black), float(white))
    return out


# ------------------------------------------------------------------------------------------------
# The reference itself (oracle/_ref), run as a child process on a box with an OpenCL GPU.
# ------------------------------------------------------------------------------------------------

def ref_available():
    """True when oracle/_ref/ref_runner exists and an OpenCL GPU device answers."""
    if not os.path.exists(REF_RUNNER):
        return False
    try:
        out = subprocess.run(["clinfo"], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return False
    for line in out.splitlines():
        if "Number of devices" in line and line.split()[-1].isdigit() and int(line.split()[-1]) > 0:
            return True
    return False


class RefSession:
    """Builds a ref_runner script; run() executes it and returns (json_lines, {name: ndarray})."""

    def __init__(self, hdr, H, W, in_stride=0, out_stride=0, delta=8, neighbor=6, black=0.0, white=255.0,
                 max_calc_res=270, workdir=None):
        self.g = make_geom(hdr, H, W, in_stride, out_stride, max_calc_res)
        self.tmp = workdir or tempfile.mkdtemp(prefix="hfref_")
        self.lines = [f"create {int(hdr)} {H} {W} {in_stride} {out_stride} {delta} {neighbor} {black} {white} {max_calc_res}"]
        self.outputs = {}
        self.nfiles = 0

    def _file(self, tag):
        self.nfiles += 1
        return os.path.join(self.tmp, f"{self.nfiles:04d}_{tag}.bin")

    def radius(self, R): self.lines.append(f"radius {R}")
    def params(self, delta, neighbor, black, white): self.lines.append(f"params {delta} {neighbor} {black} {white}")
    def framecount(self, n): self.lines.append(f"framecount {n}")

    def update(self, frame):
        p = self._file("in")
        np.ascontiguousarray(frame).tofile(p)
        self.lines.append(f"update {p}")

    def calc(self): self.lines.append("calc")
    def warp(self, t, mode): self.lines.append(f"warp {float(t)!r} {mode}")
    def copy(self): self.lines.append("copy")
    def stats(self): self.lines.append("stats")
    def time_calc(self, n): self.lines.append(f"time_calc {n}")
    def time_warp(self, n, t, mode): self.lines.append(f"time_warp {n} {float(t)!r} {mode}")

    def download(self, name):
        p = self._file("out"); self.outputs[name] = (p, frame_dtype(self.g.hdr), None)
        self.lines.append(f"download {p}")

    def dump_offsets(self, name):
        p = self._file("off"); self.outputs[name] = (p, np.int16, (2, self.g.lh, self.g.lw))
        self.lines.append(f"dump_offsets {p}")

    def dump_blurred(self, idx, name):
        p = self._file("blur"); self.outputs[name] = (p, np.int16, (2, self.g.lh, self.g.lw))
        self.lines.append(f"dump_blurred {idx} {p}")

    def run(self, timeout=600):
        script = os.path.join(self.tmp, "script.txt")
        with open(script, "w") as f:
            f.write("\n".join(self.lines) + "\n")
        r = subprocess.run([REF_RUNNER, script], capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError(f"ref_runner rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
        js = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        arrs = {}
        for name, (p, dt, shape) in self.outputs.items():
            a = np.fromfile(p, dtype=dt)
            arrs[name] = a.reshape(shape) if shape else a
        return js, arrs
