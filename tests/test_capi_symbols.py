"""CPU: the C-ABI library builds, loads, and exports every symbol include/hopperflow.h and include/hopperflow_diag.h declare; the ctypes
table covers them all; no compute call is made (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "hopperflow.h")).read() + open(os.path.join(ROOT, "include", "hopperflow_diag.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hf_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(native_lib):
    from hopperrender_amd import build, capi
    syms = declared_symbols()
    assert len(syms) >= 30
    nm = subprocess.run(["nm", "-D", "--defined-only", build.LIB_FLOW], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (hf_[a-z0-9_]+)", nm))
    assert not (set(syms) - exported), f"declared but not exported: {sorted(set(syms) - exported)}"
    assert not (set(syms) - set(capi.SIGNATURES)), f"not bound in capi.py: {sorted(set(syms) - set(capi.SIGNATURES))}"
    for s in syms:
        assert getattr(native_lib, s) is not None
    assert native_lib.hf_abi_version() == 6   # round 6: hopperflow_diag.h (hf_debug_counters_*, hf_timeline_record.duration_ms); round 5: hf_batch_timeline_enable / _read, hf_clock_probe (round 4: hf_debug_bounds_*; round 3: hf_batch_run_period / hf_batch_sync, hf_select_device, device_index = -1)


def test_struct_layouts_match_the_header(native_lib, tmp_path):
    """sizeof of the C structs as compiled by gcc == the ctypes mirrors."""
    from hopperrender_amd import capi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "hopperflow.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(hf_config), sizeof(hf_params), sizeof(hf_stats), sizeof(hf_profile), sizeof(hf_filter_config), sizeof(hf_filter_state));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [ctypes.sizeof(capi.HfConfig), ctypes.sizeof(capi.HfParams), ctypes.sizeof(capi.HfStats), ctypes.sizeof(capi.HfProfile),
                     ctypes.sizeof(capi.HfFilterConfig), ctypes.sizeof(capi.HfFilterState)]


def test_first_suitable_device_rule(native_lib):
    """detectDevices (reference opticalFlowCalc.cpp:67-109) on a fake capability table: the FIRST device with enough memory,
    >= 2 KB of LDS and 16 x 16 workgroups wins (hf_create(device_index = -1) applies the same function to the HIP devices);
    when none qualifies the message names what the last one lacked, in the reference's words."""
    from hopperrender_amd import capi
    GB = 1 << 30
    caps = lambda rows: (capi.HfDeviceCaps * len(rows))(*[capi.HfDeviceCaps(*r) for r in rows])
    why = ctypes.create_string_buffer(384)
    sel = lambda rows, need: native_lib.hf_select_device(caps(rows), len(rows), need, why, len(why))
    good, small, no_lds, tiny_wg, wave32 = (288 * GB, 65536, 1024, 64), (1 * GB, 65536, 1024, 64), (288 * GB, 1024, 1024, 64), (288 * GB, 65536, 128, 64), (288 * GB, 65536, 1024, 32)
    assert sel([good, good], 4 * GB) == 0                          # first suitable, not "best"
    assert sel([small, good, good], 4 * GB) == 1                   # device 0 lacks the memory: moves on (the reference's loop, :75-93)
    assert sel([no_lds, tiny_wg, wave32, small, good], 4 * GB) == 4
    assert sel([small], 1 * GB) == 0                               # >= , not >
    assert sel([good, small], 4 * GB) == 0
    assert sel([small, no_lds], 4 * GB) == -1
    assert b"Not enough shared memory available! Required: 2048 bytes, Available: 1024 bytes" in why.value   # the LAST device inspected (:98-108)
    assert sel([no_lds, small], 4 * GB) == -1
    assert b"Not enough VRAM available! Required: 4096 MB, Available: 1024 MB" in why.value
    assert sel([tiny_wg], 1) == -1 and b"work group sizes" in why.value
    assert sel([wave32], 1) == -1 and b"Wavefront size 32" in why.value
    assert native_lib.hf_select_device(None, 0, 1, None, 0) == -1


def test_no_device_fails_loudly_not_silently(native_lib):
    """Without a GPU the product must raise (there is no CPU fallback); with one it must construct."""
    import pytest
    from hopperrender_amd.calc import OpticalFlowCalcSDR
    from hopperrender_amd.capi import HopperFlowError
    if native_lib.hf_device_count() == 0:
        with pytest.raises(HopperFlowError, match="detectDevices"):
            OpticalFlowCalcSDR(180, 320)
    else:
        OpticalFlowCalcSDR(180, 320).close()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under hopperrender_amd/ or include/ may reference it."""
    bad = []
    for d in ("hopperrender_amd", "include"):
        for root, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(root, f), errors="replace").read()
                    if re.search(r"(from|import)\s+oracle|hf_oracle|libhf_oracle|oracle/", txt):
                        bad.append(os.path.join(root, f))
    assert not bad, bad


def test_cpp_adapter_header_is_self_contained(tmp_path):
    """include/opticalFlowCalc.h compiles with plain g++ (no HIP, no OpenCL, no Windows headers)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "opticalFlowCalc.h"\nint main(){ OpticalFlowCalc* p = nullptr; (void)p; return sizeof(OpticalFlowCalcSDR) > 0 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), str(src)])


def test_bench_names_kernels_the_library_contains():
    """bench.py prints roofline.kernel for rocprofv3 --kernel-trace to be matched against (the string had drifted from the
    shipped symbol after a template parameter was added).  The name is read from the binary; exactly one kernel matches each prefix."""
    import bench
    from hopperrender_amd import build, capi
    build.build_all()
    for (hdr, big), prefix in bench.WARP_SYMBOL_PREFIX.items():
        name = bench.warp_symbol(hdr, *((2160, 3840) if big else (1080, 1920)))
        assert name.startswith(prefix) and name.endswith(">") and "(" not in name
        assert name in subprocess.run(["nm", "-C", capi.lib_path()], capture_output=True, text=True, check=True).stdout
    with pytest.raises(RuntimeError):
        capi.kernel_symbol("no_such_kernel<")
    with pytest.raises(RuntimeError):
        capi.kernel_symbol("warp_wg_kernel<")          # ambiguous: several instantiations
    assert set(bench.OTHER_WORKLOADS) < set(bench.WORKLOADS) and set(bench.CONTENT_LEGS) < set(bench.WORKLOADS)
