"""Builds the native libraries IN-TREE (so the .so files travel with the repo snapshot).

    libhopperflow.so       HIP kernels + C ABI (include/hopperflow.h)        hipcc, gfx950 only
    libopticalflowcalc.so  source-compatible C++ adapter (include/opticalFlowCalc.h)   g++

`python -m hopperrender_amd.build` or hopperrender_amd.build.build_all().  hipcc cross-compiles
without a GPU.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
INCLUDE = os.path.join(ROOT, "include")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
             "-Wall", "-Wno-unused-function"]

FLOW_SOURCES = ("hf_kernels.hip", "hf_flow.hip", "hf_context.hip", "hf_calc.hip", "hf_batch.hip", "hf_async_io.hip", "hf_filter.cpp", "hf_hostio.cpp")
FLOW_HEADERS = ("hf_kernels.h", "hf_phase_plane.h", "hf_ctx.h")
LIB_FLOW = os.path.join(LIBDIR, "libhopperflow.so")
LIB_FLOW_DEBUG = os.path.join(LIBDIR, "libhopperflow_dbg.so")   # --debug-bounds: -DHF_DEBUG_BOUNDS, every gather index checked on the device
LIB_ADAPTER = os.path.join(LIBDIR, "libopticalflowcalc.so")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout[-4000:] + r.stderr[-8000:])
        raise RuntimeError("native build failed")
    return r


def build_flow(force=False, debug_bounds=False):
    """libhopperflow.so, or -- debug_bounds -- libhopperflow_dbg.so: the same sources with -DHF_DEBUG_BOUNDS (csrc/hf_kernels.h
    HF_DBG_CHECK: every gather index of the kernels checked against its buffer, trap on violation).  Select it with HF_LIB=<path>."""
    os.makedirs(LIBDIR, exist_ok=True)
    extra = os.environ.get("HF_CXXFLAGS", "").split()   # experiments only (e.g. -DHF_EXP=1)
    force = force or bool(extra)
    target, suffix = (LIB_FLOW_DEBUG, ".dbg.o") if debug_bounds else (LIB_FLOW, ".o")
    if debug_bounds:
        extra = extra + ["-DHF_DEBUG_BOUNDS"]
    srcs = [os.path.join(CSRC, f) for f in FLOW_SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in FLOW_HEADERS] + [os.path.join(INCLUDE, "hopperflow.h"), os.path.join(INCLUDE, "hopperflow_diag.h"), os.path.join(INCLUDE, "config.h")]
    if force or _stale(target, deps):
        objs = []
        hdrs = deps[len(srcs):]
        todo = []
        for s in srcs:
            o = os.path.join(LIBDIR, os.path.basename(s) + suffix)
            if force or _stale(o, [s] + hdrs):   # objects are git-ignored; only changed sources are recompiled
                todo.append([HIPCC] + HIP_FLAGS + extra + ["-I", INCLUDE, "-c", s, "-o", o])
            objs.append(o)
        if todo:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(min(4, len(todo))) as ex:   # (the translation units are independent; hipcc is single-threaded)
                list(ex.map(_run, todo))
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs +
             ["-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"])
    return target


def build_adapter(force=False):
    src = os.path.join(CSRC, "opticalFlowCalc.cpp")
    if not os.path.exists(src):
        return None
    deps = [src, os.path.join(INCLUDE, "opticalFlowCalc.h"), os.path.join(INCLUDE, "hopperflow.h"), os.path.join(INCLUDE, "hopperflow_diag.h"), os.path.join(INCLUDE, "config.h")]
    if force or _stale(LIB_ADAPTER, deps + [LIB_FLOW]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", INCLUDE, src, "-o", LIB_ADAPTER,
              "-L", LIBDIR, "-lhopperflow", "-Wl,-rpath,$ORIGIN", "-Wl,--no-undefined"])
    return LIB_ADAPTER


def build_all(force=False):
    build_flow(force)
    build_adapter(force)
    return LIB_FLOW


if __name__ == "__main__":
    if "--debug-bounds" in sys.argv:
        print(build_flow(force="--force" in sys.argv, debug_bounds=True))
    else:
        build_all(force="--force" in sys.argv)
        print(LIB_FLOW)
