/* include/hopperflow_diag.h -- measurement and debugging aids of libhopperflow.so.
 *
 * NOT part of the drop-in boundary: nothing here replaces a member of the reference's OpticalFlowCalc (opticalFlowCalc.h:100-132), and a
 * host that binds the path (cgo / JNI / N-API / ctypes: INTEGRATION.md) binds include/hopperflow.h only.  These entry points exist for the
 * build's own tests, bench.py and the tools under tools/: per-dispatch timestamps of a batch without a profiler, device-side counters of the
 * decisions the kernels take per window / workgroup, the bounds-checking build's records, clock / HBM probes of the box, the device's
 * v_rcp_f32 for parity tooling, and HIP-event timers on a context's stream.  Same library, same error convention (0 = ok, hf_last_error).
 */
#ifndef HOPPERFLOW_DIAG_H
#define HOPPERFLOW_DIAG_H

#include "hopperflow.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Timeline of a batch WITHOUT a profiler: while it is on, every dispatch hf_batch_run_period issues (grid samples, fused period warp, each
 * launch of the refinement chain -- issued one by one instead of as a graph replay -- and the blur) carries the start / stop events of
 * the dispatch itself (hipExtLaunchKernelGGL), i.e. the timestamps a kernel trace would read, on the clock of the device and relative to
 * ONE reference per process and device, so the records of several batches (streams) line up.  (rocprofv3's kernel trace costs enough per
 * dispatch to make four batch streams host-bound: its timeline is not the un-profiled run's.)  Checked against that trace inside ONE
 * profiled process, dispatch by dispatch (tools/timeline_vs_trace.py, profiles/r06_timeline_vs_trace_*.txt): the stop event is the
 * dispatch's completion time within 1.4 us over 1,036 dispatches per stream; the start event lies 3.8-4.5 us BEFORE the dispatch's own start
 * timestamp (the runtime stamps it when the packet reaches the head of its queue), so duration_ms = the kernel's run time + about 4 us
 * of hand-over -- longer only for the first launch of a period behind an idle queue.  The first skip_periods calls of
 * hf_batch_run_period after _enable pass unobserved (so a driver can arm the timeline before its timed region -- _enable synchronises the
 * batch's stream -- and have it record in the middle); it switches itself off when fewer than 32 of the max_launches records are left (a
 * period needs about 16; max_launches must be 0 or >= 32: HF_ERR_INVALID_ARGUMENT otherwise).  hf_batch_timeline_enable(batch, 0, 0)
 * switches it off and frees the events; _read synchronises the batch's stream and the warp streams of HF_FLAG_DUAL_STREAM members and
 * returns the records taken so far (*n_records = how many exist; at most `capacity` are written; a record whose events are not ready is
 * flagged, not fatal); hf_batch_timeline_dropped = launches issued while on that found no free record. */
typedef struct hf_timeline_record {
    char kernel[32];      /* "grid_samples", "warp_period", "plane", "large_windows_x" / "_y", "level_32" ... "level_2", "blur" */
    int32_t period;       /* hf_batch_run_period calls since the recording started */
    int32_t flags;        /* bit 0: the dispatch's events were not ready / not recorded (its launch failed): times are 0 */
    double start_ms;      /* start / end of the dispatch, milliseconds since the process's reference event on this device (float32
                           * resolution of a long elapsed time: ~0.25 us at 2-4 s, ~2 us at 16-32 s -- take durations from duration_ms) */
    double end_ms;
    double duration_ms;   /* end - start of THIS dispatch, measured directly between its own two events */
} hf_timeline_record;
int hf_batch_timeline_enable(hf_batch* batch, int max_launches, int skip_periods);
int hf_batch_timeline_read(hf_batch* batch, hf_timeline_record* out, int capacity, int* n_records);
uint64_t hf_batch_timeline_dropped(const hf_batch* batch);

/* Device-side counters of what the kernels decide, per context (a batch counts in its first member): which path the workgroups of the fused
 * period warp took -- staged LDS window / interior global path (window too large: fast or diverging motion) / generic body (tile edge,
 * mirror zone) -- and, per refinement level, how many windows of the table tiles there were and how many of them summed their blocks' SAD
 * vectors instead of gathering the phase plane again (csrc/hf_flow.hip "SAD TABLES").  Content-dependent: bench.py prints them per scene.
 * _enable(ctx, 1) allocates and zeroes them and drops the context's (and its batch's) cached graphs; while on, every counted wave /
 * workgroup issues one or two atomics.  _read synchronises the context. */
typedef struct hf_debug_counters {
    uint32_t warp_workgroups[3];     /* staged, interior-global, generic */
    uint32_t reserved;
    uint32_t level_windows[16][2];   /* [level k][axis]: windows of full tiles at the full search radius */
    uint32_t level_reused[16][2];    /* ... of those: reused */
    int32_t level_window_size[16];   /* window size of level k of the last chain (0: no such level) */
} hf_debug_counters;
int hf_debug_counters_enable(hf_ctx* ctx, int on);
int hf_debug_counters_read(hf_ctx* ctx, hf_debug_counters* out, int reset);

/* Device debug build (`python -m hopperrender_amd.build --debug-bounds` -> libhopperflow_dbg.so, -DHF_DEBUG_BOUNDS): every gather index of
 * the kernels -- frame, phase-plane, flow-table and LDS-window reads -- is checked against its buffer and violations are recorded on the
 * device (the only device-side memory check there can be where GPU AddressSanitizer is unavailable; the reference's own out-of-range
 * case is the single reflection of calcDeltaSumsKernelSDR.h:86-95).  hf_debug_bounds_violations synchronises the device and returns the
 * number of violations since the last reset and site / block / thread / source line of the first; hf_debug_bounds_selftest issues 64
 * out-of-range indices (site 999) and checks that exactly those were recorded.  Both return HF_ERR_STATE in the product build, which
 * compiles the checks away. */
int hf_debug_bounds_violations(hf_ctx* ctx, uint32_t* count, uint32_t first[4], int reset);
int hf_debug_bounds_selftest(hf_ctx* ctx);
/* v_rcp_f32 of the device for n <= 32 values.  The reference's apply_levels* divide through it when
 * built by AMD OpenCL (x / y -> x * rcp(y)); CPU checkers use this to reproduce levels bit-exactly. */
int hf_device_rcp(hf_ctx* ctx, const float* host_in, float* host_out, int n);

/* Clock of the shader array RIGHT NOW, in MHz: one wave compares the shader-cycle counter with the 100 MHz reference counter over
 * duration_us microseconds, on a stream of its own, while whatever else the process has queued keeps running (blocks until the probe
 * has run).  The chip lowers its clock under load by a device-dependent amount; bench.py samples this behind its last warm-up step (the same load, outside the timed region)
 * so that lines from different boxes can be normalised. */
int hf_clock_probe(int device_index, int duration_us, double* shader_mhz);
/* What this device's HBM sustains for a plain streaming copy of `bytes` bytes (16 bytes per lane, non-temporal loads and stores; best of
 * `repeats` passes; read + write bytes per second, GB/s): the yardstick a bandwidth-bound pipeline should be held against on THIS box. */
int hf_hbm_copy_probe(int device_index, size_t bytes, int repeats, double* read_plus_write_GBps);

/* ---- measurement: HIP events on ctx's own stream (torch events cannot see this stream) ---- */
int hf_timer_begin(hf_ctx* ctx);
int hf_timer_end(hf_ctx* ctx, float* elapsed_ms); /* synchronises on the end event */

#ifdef __cplusplus
}
#endif
#endif
