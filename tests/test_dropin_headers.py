"""CPU: the C++ drop-in surface compiles behind the reference filter's own #include lines (HopperRender.cpp:24-25)
with plain g++, exposes the config.h macros transitively (opticalFlowCalc.h:8) and leaks no `max` macro."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def test_dropin_headers_compile_like_the_filter(tmp_path):
    obj = str(tmp_path / "dropin.o")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", INC, "-c",
                        os.path.join(ROOT, "tests", "cpp", "dropin_headers.cpp"), "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_forwarders_and_config_exist():
    for h in ("opticalFlowCalc.h", "opticalFlowCalcSDR.h", "opticalFlowCalcHDR.h", "config.h", "hopperflow.h"):
        assert os.path.exists(os.path.join(INC, h)), h
    # the filter's own config.h wins when it precedes ours on the include path: every macro is #ifndef-guarded
    txt = open(os.path.join(INC, "config.h")).read()
    for m in ("MAX_CALC_RES", "NUM_ITERATIONS", "MIN_SEARCH_RADIUS", "MAX_SEARCH_RADIUS", "UPPER_PERF_BUFFER", "LOWER_PERF_BUFFER",
              "CALC_TIME_INTERVAL", "DEFAULT_DELTA_SCALAR", "DEFAULT_NEIGHBOR_SCALAR", "DEFAULT_BLACK_LEVEL", "DEFAULT_WHITE_LEVEL",
              "DEFAULT_SCENE_CHANGE_THRESHOLD", "DEFAULT_BUFFER_FRAMES"):
        assert f"#ifndef {m}\n#define {m} " in txt, m


def test_replay_and_ownership_programs_compile(tmp_path):
    for src in ("replay_filter.cpp", "field_ownership.cpp"):
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I", INC, "-c", os.path.join(ROOT, "tests", "cpp", src),
                            "-o", str(tmp_path / (src + ".o"))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
