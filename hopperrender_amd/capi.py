"""ctypes binding of the C ABI (include/hopperflow.h + the diagnostics of include/hopperflow_diag.h) -- the same symbols a cgo/JNI/N-API host binds.

There is NO fallback: if libhopperflow.so is missing or cannot be loaded, importing the product
path raises (the library is built in-tree by hopperrender_amd.build / __graft_entry__.build()).
"""
import ctypes as C
import os

from . import build as _build

HF_FLAG_ASYNC = 0x1
HF_FLAG_NO_GRAPH = 0x2
HF_FLAG_PROFILE = 0x4
HF_FLAG_NO_LAZY_ARGMIN = 0x8
HF_FLAG_DUAL_STREAM = 0x40
HF_FLAG_NO_FUSED_WARP = 0x80
HF_FLAG_NO_TIMING = 0x200
HF_FLAG_BATCH_NORMAL_PRIORITY = 0x800
HF_FLAG_BATCH_EAGER_PLANES = 0x1000
HF_FLAG_NO_SAD_REUSE = 0x2000
HF_FLAG_SAD_REUSE_ALWAYS = 0x4000
HF_MAX_PERIOD_OUTPUTS = 6

(HF_OK, HF_ERR_INVALID_ARGUMENT, HF_ERR_NO_DEVICE, HF_ERR_OUT_OF_MEMORY, HF_ERR_HIP, HF_ERR_STATE) = (0, -1, -2, -3, -4, -5)


class HfConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("is_hdr", C.c_int32), ("frame_height", C.c_int32),
                ("frame_width", C.c_int32), ("input_stride", C.c_int32), ("output_stride", C.c_int32),
                ("delta_scalar", C.c_int32), ("neighbor_scalar", C.c_int32), ("black_level", C.c_float),
                ("white_level", C.c_float), ("max_calc_res", C.c_int32), ("device_index", C.c_int32),
                ("iterations", C.c_int32), ("blur_radius", C.c_int32), ("search_radius", C.c_int32),
                ("flags", C.c_uint32)]


class HfDeviceCaps(C.Structure):
    _fields_ = [("vram_bytes", C.c_uint64), ("lds_bytes_per_workgroup", C.c_uint64), ("max_threads_per_workgroup", C.c_int32),
                ("wavefront_size", C.c_int32)]


class HfTimelineChunk(C.Structure):
    _fields_ = [("first_period", C.c_int64), ("n_periods", C.c_int64), ("first_frame", C.c_int64), ("n_frames", C.c_int64),
                ("first_output", C.c_int64), ("n_outputs", C.c_int64), ("blend_at_start", C.c_double)]


class HfHostioConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("in_ring", C.c_int32), ("out_ring", C.c_int32), ("frame_output_mode", C.c_int32),
                ("scene_change_threshold", C.c_int32), ("reserved", C.c_int32), ("source_frame_time", C.c_int64),
                ("target_frame_time", C.c_int64)]


HF_HOSTIO_FILL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p)
HF_HOSTIO_SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32)


class HfParams(C.Structure):
    _fields_ = [("delta_scalar", C.c_int32), ("neighbor_scalar", C.c_int32), ("black_level", C.c_float),
                ("white_level", C.c_float), ("search_radius", C.c_int32), ("frame_count", C.c_uint32)]


class HfStats(C.Structure):
    _fields_ = [("total_frame_delta", C.c_uint32), ("frame_count", C.c_uint32), ("ofc_calc_time", C.c_double),
                ("ofc_avg_calc_time", C.c_double), ("ofc_peak_calc_time", C.c_double), ("warp_calc_time", C.c_double),
                ("res_scalar", C.c_int32), ("low_width", C.c_int32), ("low_height", C.c_int32),
                ("frame_width", C.c_int32), ("frame_height", C.c_int32), ("input_stride", C.c_int32),
                ("output_stride", C.c_int32), ("iterations", C.c_int32), ("initial_window", C.c_int32),
                ("input_frame_bytes", C.c_uint64), ("output_frame_bytes", C.c_uint64), ("phase_plane_bytes", C.c_uint64),
                ("sad_tables", C.c_int32), ("still_share", C.c_float)]


class HfTimelineRecord(C.Structure):
    _fields_ = [("kernel", C.c_char * 32), ("period", C.c_int32), ("flags", C.c_int32), ("start_ms", C.c_double), ("end_ms", C.c_double),
                ("duration_ms", C.c_double)]


class HfDebugCounters(C.Structure):
    _fields_ = [("warp_workgroups", C.c_uint32 * 3), ("reserved", C.c_uint32), ("level_windows", (C.c_uint32 * 2) * 16),
                ("level_reused", (C.c_uint32 * 2) * 16), ("level_window_size", C.c_int32 * 16)]


class HfProfile(C.Structure):
    _fields_ = [("warp_launches", C.c_uint64), ("warp_ms", C.c_double), ("copy_launches", C.c_uint64),
                ("copy_ms", C.c_double), ("flow_chains", C.c_uint64), ("flow_ms", C.c_double), ("warp_frames", C.c_uint64)]


class HfFilterConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("scene_change_threshold", C.c_int32), ("source_frame_time", C.c_int64),
                ("target_frame_time", C.c_int64), ("frame_output_mode", C.c_int32), ("auto_adjust", C.c_int32),
                ("active", C.c_int32), ("reserved", C.c_int32)]


class HfFilterState(C.Structure):
    _fields_ = [("num_int_frames", C.c_int32), ("active", C.c_int32), ("blending_scalar", C.c_double),
                ("total_warp_duration", C.c_double), ("playback_frame_time", C.c_int64),
                ("peak_scene_change_delta", C.c_uint32), ("peak_scene_change_delta2", C.c_uint32),
                ("frame_delta_history", C.c_uint32), ("scene_change_history", C.c_uint32),
                ("average_frame_delta", C.c_int32), ("scene_change_delta1", C.c_int32), ("scene_change_delta2", C.c_int32)]


# name -> (restype, argtypes); must list every symbol include/hopperflow.h declares (tests check this)
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "hf_create": (_i, [C.POINTER(HfConfig), C.POINTER(_vp)]),
    "hf_destroy": (None, [_vp]),
    "hf_last_error": (C.c_char_p, [_vp]),
    "hf_abi_version": (_i, []),
    "hf_select_device": (_i, [C.POINTER(HfDeviceCaps), _i, C.c_uint64, C.c_char_p, C.c_size_t]),
    "hf_get_device": (_i, [_vp]),
    "hf_update_frame": (_i, [_vp, _vp]),
    "hf_calculate_optical_flow": (_i, [_vp]),
    "hf_warp_frames": (_i, [_vp, _f, _i]),
    "hf_copy_frame": (_i, [_vp]),
    "hf_download_frame": (_i, [_vp, _vp]),
    "hf_get_params": (_i, [_vp, C.POINTER(HfParams)]),
    "hf_set_params": (_i, [_vp, C.POINTER(HfParams)]),
    "hf_get_stats": (_i, [_vp, C.POINTER(HfStats)]),
    "hf_update_frame_device": (_i, [_vp, _vp]),
    "hf_update_frame_device_ref": (_i, [_vp, _vp]),
    "hf_update_frame_async": (_i, [_vp, _vp]),
    "hf_download_frame_async": (_i, [_vp, _vp]),
    "hf_wait_flow": (_i, [_vp]),
    "hf_downloads_issued": (C.c_uint64, [_vp]),
    "hf_wait_download": (_i, [_vp, C.c_uint64]),
    "hf_interpolate_period": (_i, [_vp, _vp, _i, C.POINTER(C.c_float), C.POINTER(_vp), _i]),
    "hf_interpolate_period_ex": (_i, [_vp, _vp, _i, C.POINTER(C.c_float), C.POINTER(_vp), _i, _i]),
    "hf_batch_create": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp)]),
    "hf_batch_destroy": (None, [_vp]),
    "hf_batch_calculate_optical_flow": (_i, [_vp]),
    "hf_batch_update_frames_device_ref": (_i, [_vp, C.POINTER(_vp)]),
    "hf_batch_interpolate_period": (_i, [_vp, C.POINTER(_i), C.POINTER(C.c_float), C.POINTER(_vp), _i]),
    "hf_batch_run_period": (_i, [_vp, C.POINTER(_vp), _i, C.POINTER(_i), C.POINTER(C.c_float), C.POINTER(_vp), _i]),
    "hf_batch_defers_planes": (_i, [_vp]),
    "hf_batch_sync": (_i, [_vp]),
    "hf_batch_size": (_i, [_vp]),
    "hf_batch_timeline_enable": (_i, [_vp, _i, _i]),
    "hf_batch_timeline_read": (_i, [_vp, C.POINTER(HfTimelineRecord), _i, C.POINTER(_i)]),
    "hf_batch_timeline_dropped": (C.c_uint64, [_vp]),
    "hf_debug_counters_enable": (_i, [_vp, _i]),
    "hf_debug_counters_read": (_i, [_vp, C.POINTER(HfDebugCounters), _i]),
    "hf_batch_last_error": (C.c_char_p, [_vp]),
    "hf_download_frame_device": (_i, [_vp, _vp]),
    "hf_set_output_buffer": (_i, [_vp, _vp]),
    "hf_sync": (_i, [_vp]),
    "hf_read_offsets": (_i, [_vp, _vp]),
    "hf_read_blurred_flow": (_i, [_vp, _i, _vp]),
    "hf_read_phase_plane": (_i, [_vp, _i, _vp, _vp]),
    "hf_write_blurred_flow": (_i, [_vp, _i, _vp]),
    "hf_device_rcp": (_i, [_vp, _vp, _vp, _i]),
    "hf_get_profile": (_i, [_vp, C.POINTER(HfProfile)]),
    "hf_reset_profile": (_i, [_vp]),
    "hf_set_profile_interval": (_i, [_vp, _i, _i]),
    "hf_timer_begin": (_i, [_vp]),
    "hf_timer_end": (_i, [_vp, C.POINTER(C.c_float)]),
    "hf_shard_timeline": (_i, [C.c_int64, _i, _i, C.c_int64, C.c_int64, _i, _i, C.POINTER(HfTimelineChunk), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int64]),
    "hf_hostio_create": (_i, [_vp, C.POINTER(HfHostioConfig), C.POINTER(_vp)]),
    "hf_hostio_destroy": (None, [_vp]),
    "hf_hostio_run": (_i, [_vp, C.POINTER(HfTimelineChunk), C.POINTER(C.c_int32), C.POINTER(C.c_float), HF_HOSTIO_FILL_FN, HF_HOSTIO_SINK_FN, _vp, C.POINTER(C.c_int32)]),
    "hf_hostio_get_traffic": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hf_hostio_last_error": (C.c_char_p, [_vp]),
    "hf_device_count": (_i, []),
    "hf_clock_probe": (_i, [_i, _i, C.POINTER(C.c_double)]),
    "hf_hbm_copy_probe": (_i, [_i, C.c_size_t, _i, C.POINTER(C.c_double)]),
    "hf_device_malloc": (_i, [_i, C.c_size_t, C.POINTER(_vp)]),
    "hf_device_free": (_i, [_i, _vp]),
    "hf_memcpy_h2d": (_i, [_i, _vp, _vp, C.c_size_t]),
    "hf_memcpy_d2h": (_i, [_i, _vp, _vp, C.c_size_t]),
    "hf_host_malloc_pinned": (_i, [C.c_size_t, C.POINTER(_vp)]),
    "hf_host_free_pinned": (_i, [_vp]),
    "hf_filter_create": (_i, [C.POINTER(HfFilterConfig), C.POINTER(_vp)]),
    "hf_debug_bounds_selftest": (_i, [_vp]),
    "hf_debug_bounds_violations": (_i, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _i]),
    "hf_filter_destroy": (None, [_vp]),
    "hf_filter_new_segment": (_i, [_vp, C.c_double]),
    "hf_filter_set_playback_frame_time": (_i, [_vp, C.c_int64]),
    "hf_filter_is_active": (_i, [_vp]),
    "hf_filter_begin_source_frame": (_i, [_vp]),
    "hf_filter_blending_scalar": (C.c_double, [_vp]),
    "hf_filter_advance_blending_scalar": (None, [_vp]),
    "hf_filter_add_warp_duration": (None, [_vp, C.c_double]),
    "hf_filter_auto_adjust": (_i, [_vp, C.c_double, C.POINTER(C.c_int32)]),
    "hf_filter_push_frame_delta": (_i, [_vp, C.c_uint32, C.c_uint32]),
    "hf_filter_detect_scene_change": (_i, [_vp, C.c_uint32]),
    "hf_filter_get_state": (_i, [_vp, C.POINTER(HfFilterState)]),
    "hf_filter_deliver": (_i, [_vp, _vp, _vp, C.POINTER(_vp), _i, C.POINTER(_i), C.POINTER(C.c_int32)]),
}

_lib = None
_devices_used = set()   # devices this process created contexts on (calc.OpticalFlowCalc); read by the debug-bounds exit check


def is_debug_bounds_build():
    return "libhopperflow_dbg" in os.path.basename(lib_path())


def debug_bounds_violations(devices=None):
    """Violation records of the bounds-checking build (csrc/hf_kernels.h HF_DBG_CHECK) of THIS process: {device: (count, first[4])} for every
    device it opened contexts on.  The records are per process, device and translation unit, so every process that does GPU work under
    libhopperflow_dbg.so has to read its own -- the exit hook below does, for test children (bench.py ranks, host-I/O workers, cli workers)."""
    L = load()
    out = {}
    for dev in sorted(devices if devices is not None else _devices_used):
        cfg = HfConfig(struct_size=C.sizeof(HfConfig), is_hdr=0, frame_height=64, frame_width=96, delta_scalar=8, neighbor_scalar=6,
                       black_level=0.0, white_level=255.0, max_calc_res=270, device_index=dev)
        ctx = _vp()
        check(L.hf_create(C.byref(cfg), C.byref(ctx)))
        n, first = C.c_uint32(0), (C.c_uint32 * 4)()
        rc = L.hf_debug_bounds_violations(ctx, C.byref(n), first, 0)
        L.hf_destroy(ctx)
        if rc == 0:
            out[dev] = (n.value, list(first))
    return out


def _debug_bounds_exit_check():
    try:
        bad = {d: v for d, v in debug_bounds_violations().items() if v[0]}
    except Exception as e:   # (a process that lost its GPU cannot report; say so, do not mask the original failure)
        import sys
        sys.stderr.write(f"[HopperRender] debug-bounds exit check could not run: {e!r}\n")
        return
    if bad:
        import sys
        sys.stderr.write(f"[HopperRender] HF_DEBUG_BOUNDS: out-of-range gather indices recorded in process {os.getpid()}: {bad} (device: (count, [site, block, thread, line]))\n")
        sys.stderr.flush()
        os._exit(97)


def lib_path():
    # HF_LIB: another build of the SAME library (kernel experiments); never an alternative implementation
    return os.environ.get("HF_LIB") or _build.LIB_FLOW


def load():
    """Load libhopperflow.so (raises if absent: the product has no CPU path)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m hopperrender_amd.build` "
                               "(or __graft_entry__.build()); there is no fallback implementation")
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        if is_debug_bounds_build():   # every process under the bounds-checking build answers for its own violation records when it ends
            import atexit
            atexit.register(_debug_bounds_exit_check)
    return _lib


def kernel_symbol(prefix, path=None):
    """The one kernel of the library whose demangled name contains `prefix`, as rocprofv3's kernel trace spells it up to the closing
    '>' of its template arguments (e.g. 'warp_wg_kernel<unsigned short, 2, 4, 2>'), read from the library's symbol table with `nm -C`
    (HIP keeps a handle symbol of that name per __global__ function).  Raises unless exactly one kernel matches."""
    import subprocess
    out = subprocess.run(["nm", "-C", path or lib_path()], capture_output=True, text=True, check=True).stdout
    found = set()
    for line in out.splitlines():
        i = line.find("::" + prefix)
        if i >= 0:
            name = line[i + 2:]
            found.add(name[:name.index("(")] if "(" in name else name)
    if len(found) != 1:
        raise RuntimeError(f"kernel symbol '{prefix}': {len(found)} matches in {path or lib_path()}: {sorted(found)[:4]}")
    return found.pop()


class HopperFlowError(RuntimeError):
    """Mirror of the reference's std::runtime_error (opticalFlowCalc.h:15-22); .code = hf_status."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def check(rc, ctx=None):
    if rc != 0:
        msg = load().hf_last_error(ctx)
        raise HopperFlowError(rc, (msg or b"").decode(errors="replace") or f"[HopperRender] error {rc}")
