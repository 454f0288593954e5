// hopperrender_amd/csrc/hf_calc.hip -- the five virtuals of ONE context on the gfx950 kernels (include/hopperflow.h): updateFrame
// (opticalFlowCalcSDR.cpp:19-29), calculateOpticalFlow (:44-139; the 16-step refinement chain + blur as a cached hipGraph), warpFrames
// (:141-168), copyFrame (:170-183), downloadFrame (:31-42), and the fused period calls built from them.  Layout of the ABI: hf_ctx.h.

#include "hf_ctx.h"

using namespace hfi;

namespace hfi {

// Enqueue the refinement chains + blur (opticalFlowCalcSDR.cpp:44-116) of n contexts with identical geometry and
// parameters as ONE set of launches on stream s (hf_kernels.h FlowBatch; n == 1: the plain call).  Capturable.
int enqueue_flow_chain(hf_ctx* const* cs, int n, hipStream_t s) {
    hf_ctx* c = cs[0];
    const hf::Geom& g = c->g;
    const int iters = effective_iterations(c);
    bool any_big = false;
    for (int k = 0; k < iters; k++) any_big |= c->levels[k].window > 32;
    // the window sums are zero on entry: zeroed at creation and re-zeroed by the blur kernel of every chain

    hf::FlowBatch a{};
    a.n = n;
    for (int i = 0; i < n; i++) {
        hf_ctx* m = cs[i];
        m->initial_window = initial_window(g.lw, g.lh);
        m->last_iterations = iters;
        hf::FlowStep& f = a.s[i];
        f.pp1 = m->pp[1];                                             // :79 frame N-1
        f.pp2 = m->pp[2];                                             // :80 frame N
        f.pl = m->pl;
        f.total_delta = m->d_total_delta;
        f.R = c->p.search_radius;
        f.delta_scalar = c->p.delta_scalar;
        f.neighbor_scalar = c->p.neighbor_scalar;
        f.delta_divisor = (uint32_t)(g.lh * g.lw * (g.hdr ? 6 : 10));  // :93 / HDR :93
        f.sadtab = m->sadtab; f.sad_nbx = m->sad_nbx; f.sad_nby = m->sad_nby;
        f.tables_base = m->tables; f.sums_base = m->sums;
        f.counters = c->counters;                                     // (a batch counts in its leader's)
        f.still_count = m->still_count;
    }
    hf::FlowLevel none{};
    hf::PendingArgmin pending[hf::kMaxFlowBatch] = {};   // large-window step whose argmin the next launch takes (hf_kernels.h)
    int step_index = 0;
    auto flush_pending = [&]() {   // explicit argmin launch for a pending step nobody can resolve lazily
        if (!pending[0].active) return;
        hf::FlowBatch b = a;
        for (int i = 0; i < n; i++) {
            hf::FlowStep& f = b.s[i];
            const hf::PendingArgmin& p = pending[i];
            f.cur = p.lvl; f.prev = p.lvl_prev; f.axis = p.axis; f.capture_delta = p.capture_delta;
            f.sums = const_cast<uint32_t*>(p.sums); f.use_neighbors = p.use_neighbors; f.pend = hf::PendingArgmin{};
            pending[i] = hf::PendingArgmin{};
        }
        hf::launch_flow_big_argmin(g, b, s);
    };
    const bool lazy = !(c->cfg.flags & HF_FLAG_NO_LAZY_ARGMIN);
    for (int k = 0; k < iters; k++) {                             // window halves every level (:110)
        const bool use_neighbors = k >= 4;                        // calcDeltaSumsKernelSDR.h:3,112
        if (use_neighbors) flush_pending();                       // a launch with a neighbour term reads other windows' entries
        const bool small = c->levels[k].window <= 32;
        for (int axis = 0; axis < (small ? 1 : 2); axis++) {
            for (int i = 0; i < n; i++) {
                hf_ctx* m = cs[i];
                hf::FlowStep& f = a.s[i];
                f.cur = m->levels[k];
                f.prev = k ? m->levels[k - 1] : none;             // :68-69: the chain starts from zero offsets
                f.prev2 = k > 1 ? m->levels[k - 2] : none;
                f.level_index = k;
                // SAD tables: written by every small level that has a successor's worth of blocks (windows 32 .. 4), read by every small
                // level behind a small level
                f.sad_write = c->tab_mode && m->sadtab && small && m->levels[k].window >= 4;
                f.sad_read = c->tab_mode && m->sadtab && small && k > 0 && m->levels[k - 1].window <= 32;
                f.use_neighbors = use_neighbors;
                f.axis = axis;
                f.capture_delta = (k == 0 && axis == 0);          // :91
                f.pend = pending[i];
                if (!small) f.sums = m->sums + (size_t)step_index * m->sums_stride;
                pending[i] = hf::PendingArgmin{};
            }
            if (small) {
                hf::launch_flow_level_small(g, a, s);
            } else {
                hf::launch_flow_big_partial(g, a, s);
                for (int i = 0; i < n; i++) {
                    const hf::FlowStep& f = a.s[i];
                    pending[i].active = 1; pending[i].axis = axis; pending[i].capture_delta = f.capture_delta;
                    pending[i].use_neighbors = f.use_neighbors;
                    pending[i].lvl = f.cur; pending[i].lvl_prev = f.prev; pending[i].sums = f.sums;
                }
                step_index++;
                if (!lazy || use_neighbors) flush_pending();
            }
        }
    }
    flush_pending();
    hf::BlurBatch bb{};
    bb.n = n;
    for (int i = 0; i < n; i++) {
        hf_ctx* m = cs[i];
        m->last_level = iters ? m->levels[iters - 1] : none;
        bb.s[i].last = m->last_level;
        bb.s[i].blurred = m->blurred[0];
        bb.s[i].packed = m->blurred_xy[0];
        bb.s[i].zero = any_big ? m->sums : nullptr;
        bb.s[i].still_count = m->still_count;
        bb.s[i].still_out = m->still_count ? m->d_total_delta + 1 : nullptr;
    }
    hf::launch_blur_flow(g, bb, c->cfg.blur_radius, (int)(c->sums_bytes / sizeof(uint32_t)), s);  // :115-116
    HF_HIP(c, hipGetLastError());
    return HF_OK;
}

int enqueue_flow_chain(hf_ctx* c) { return enqueue_flow_chain(&c, 1, c->stream); }

// SAD tables or not for the NEXT chain of these contexts (one decision for a batch: its launches are shared).  The tables pay when most
// windows keep their offsets from level to level (tests/flow_reuse_model.py: 74-96 % on the bench scene) and cost 10-15 % of the pipeline
// when hardly any does (every 16 x 16 block its own motion, a hard cut).  Content is coherent in time, so the chain's last kernel reports how
// many windows of the 32-level chose d = 0 on both axes -- what the next level's reuse depends on -- and the host reads whatever report has
// arrived (mapped memory, no synchronisation: one or two periods old) when it issues the next chain.  Smoothed over chains (a cut every few
// periods does not flip it), with hysteresis.  Results are identical either way; only the kernels differ.  HF_FLAG_SAD_REUSE_ALWAYS pins it on.
bool choose_tab_mode(hf_ctx* const* cs, int n) {
    hf_ctx* l = cs[0];
    if (!l->sadtab) return false;
    if (l->cfg.flags & HF_FLAG_SAD_REUSE_ALWAYS) return true;
    // A lone stream (or two, three) leaves most of the device idle: a launch is as long as its slowest wave whatever the others skip, and the
    // table kernels' computing waves take two rounds of 8 candidates per axis.  Chain alone, tables on / off: 73.0 / 70.1 us (1 pair),
    // 84.6 / 82.0 (2), 107.7 / 112.4 (4), 215.8 / 238.1 (16).
    if (n < 4) return false;
    float sum = 0.f;
    int have = 0;
    for (int i = 0; i < n; i++) {
        hf_ctx* m = cs[i];
        const uint32_t raw = m->h_total_delta ? ((volatile uint32_t*)m->h_total_delta)[1] : 0xFFFFFFFFu;
        int n32 = 0;
        for (const hf::FlowLevel& L : m->levels) if (L.window == 32) n32 = L.nwx * L.nwy;
        if (raw != 0xFFFFFFFFu && n32 > 0) {
            const float share = (float)raw / (float)n32;
            m->still_share = m->still_share < 0.f ? share : 0.5f * m->still_share + 0.5f * share;
        }
        if (m->still_share >= 0.f) { sum += m->still_share; have++; }
    }
    if (!have) return l->tab_mode;
    const float share = sum / (float)have;
    return l->tab_mode ? share >= 0.30f : share >= 0.45f;
}

// Deferred phase planes (hf_batch_run_period): the chain reads the FULL plane of frame N-1.  Members whose pp[1] still holds only
// the grid samples -- no warp launch took the build up -- get it from the stand-alone plane kernel now, in one launch.
int ensure_older_planes(hf_ctx* const* cs, int n, hipStream_t s) {
    hf::PrepBatch pb{};
    for (int i = 0; i < n; i++) {
        hf_ctx* m = cs[i];
        if (!m->plane_pending[1]) continue;
        pb.frame[pb.n] = m->ring[1]; pb.pp[pb.n] = m->pp[1]; pb.n++;
    }
    if (pb.n) {
        hf::launch_prep_frames(cs[0]->g, cs[0]->pl, pb, s);
        if (hipGetLastError() != hipSuccess) return HF_ERR_HIP;
        for (int i = 0; i < n; i++) cs[i]->plane_pending[1] = false;
    }
    return HF_OK;
}

void finish_flow_timing(hf_ctx* c) {
    // opticalFlowCalcSDR.cpp:125-138
    if (!c->flow_timing_pending) return;
    c->flow_timing_pending = false;
    float ms = 0.f;
    if (c->upload_recorded && hipEventElapsedTime(&ms, c->ev_upload, c->ev_flow_end) == hipSuccess)
        c->ofc_calc_time = (double)ms / 1e3;
    if (c->ofc_count >= kCalcTimeInterval) {
        c->ofc_avg = c->ofc_sum / c->ofc_count;
        c->ofc_count = 0;
        c->ofc_sum = 0.0;
        c->ofc_peak = c->ofc_calc_time;
    }
    c->ofc_count++;
    c->ofc_sum += c->ofc_calc_time;
    if (c->ofc_calc_time > c->ofc_peak) c->ofc_peak = c->ofc_calc_time;
}

// Warp launches go to c->warp_stream (HF_FLAG_DUAL_STREAM: a second stream).  The two streams are tied together by
// events: the warp stream waits for what the warps read, and leave_warp_stream() makes c->stream wait for the warps
// again, so every other call keeps its plain in-order semantics.
int enter_warp_stream(hf_ctx* c) {
    if (c->warp_stream == c->stream || c->on_warp_stream) return HF_OK;
    if (c->dual() && c->ev_flow_valid[0]) {
        // warpFrames reads frames N-2/N-1 and the PREVIOUS flow (blurred[0]); the chain that may have just been
        // enqueued on c->stream writes the OTHER flow buffer, so the warps only wait for the chain that produced
        // blurred[0] (recorded behind the uploads of both frames) and run side by side with the current one
        HF_HIP(c, hipStreamWaitEvent(c->warp_stream, c->ev_flow[0], 0));
    } else {
        // no tagged flow yet (the filter warps as soon as m_frameCount >= 3, before a second flow calculation):
        // order the warps behind everything enqueued so far, uploads included
        HF_HIP(c, hipEventRecord(c->ev_chain_done, c->stream));
        HF_HIP(c, hipStreamWaitEvent(c->warp_stream, c->ev_chain_done, 0));
    }
    c->on_warp_stream = true;
    return HF_OK;
}

int leave_warp_stream(hf_ctx* c) {
    if (!c->on_warp_stream) return HF_OK;
    // ev_warps_done was recorded right behind this context's last warp launch
    HF_HIP(c, hipStreamWaitEvent(c->stream, c->ev_warps_done, 0));
    c->on_warp_stream = false;
    return HF_OK;
}

int rotate_after_upload(hf_ctx* c) {
    // opticalFlowCalcSDR.cpp:22-28 : [0] <- [1] <- [2] <- new ; frame_count++
    void* f = c->ring[0];
    void* fs = c->ring_store[0];
    c->ring_store[0] = c->ring_store[1]; c->ring_store[1] = c->ring_store[2]; c->ring_store[2] = fs;
    hipEvent_t es = c->ev_slot_prep[0];
    c->ev_slot_prep[0] = c->ev_slot_prep[1]; c->ev_slot_prep[1] = c->ev_slot_prep[2]; c->ev_slot_prep[2] = es;
    uint32_t* pp = c->pp[0];
    const bool pend = c->plane_pending[0];
    c->ring[0] = c->ring[1]; c->pp[0] = c->pp[1]; c->plane_pending[0] = c->plane_pending[1];
    c->ring[1] = c->ring[2]; c->pp[1] = c->pp[2]; c->plane_pending[1] = c->plane_pending[2];
    c->ring[2] = f;          c->pp[2] = pp;       c->plane_pending[2] = pend;
    c->ring_phase = (c->ring_phase + 1) % 3;
    c->p.frame_count++;
    return HF_OK;
}

// by_reference: the ring slot points at the caller's device frame instead of receiving a copy
int update_common(hf_ctx* c, const void* src, hipMemcpyKind kind, bool by_reference) {
    if (int rc = set_device(c)) return rc;
    if (int rc = leave_warp_stream(c)) return rc;
    if (c->timing()) {
        HF_HIP(c, hipEventRecord(c->ev_upload, c->stream));  // m_ofcStartedEvent (:20)
        c->upload_recorded = true;
    }
    if (by_reference) {
        c->ring[0] = const_cast<void*>(src);
    } else {
        c->ring[0] = c->ring_store[0];
        HF_HIP(c, hipMemcpyAsync(c->ring[0], src, c->in_bytes, kind, c->stream));
    }
    hf::launch_prep_frame(c->g, c->pl, c->ring[0], c->pp[0], c->stream);
    c->plane_pending[0] = false;
    HF_HIP(c, hipGetLastError());
    if (c->io_in) HF_HIP(c, hipEventRecord(c->ev_slot_prep[0], c->stream));
    rotate_after_upload(c);
    if (!c->async()) return sync_ctx(c);
    return HF_OK;
}

int check_flow_params(hf_ctx* c) {
    const int R = c->p.search_radius;
    if (R < 2 || R > kMaxSearchRadius) return fail(c, HF_ERR_INVALID_ARGUMENT, "calculateOpticalFlow: search radius %d outside [2, 16]", R);
    if (c->p.delta_scalar < 0 || c->p.delta_scalar > 24 || c->p.neighbor_scalar < 0 || c->p.neighbor_scalar > 24)
        return fail(c, HF_ERR_INVALID_ARGUMENT, "calculateOpticalFlow: delta/neighbor scalar outside [0, 24]");
    return HF_OK;
}

// Bookkeeping behind an enqueued chain (eager, graph replay or batch): timing event, flow buffer swap.
int after_flow_enqueued(hf_ctx* c, hipStream_t s) {
    // a graph replay / batch leader skips enqueue_flow_chain()'s bookkeeping for this context: redo it
    const int iters = effective_iterations(c);
    c->initial_window = initial_window(c->g.lw, c->g.lh);
    c->last_iterations = iters;
    c->last_level = iters ? c->levels[iters - 1] : hf::FlowLevel{};
    if (c->timing()) {
        HF_HIP(c, hipEventRecord(c->ev_flow_end, s));
        c->flow_timing_pending = true;
    }
    c->delta_pending = c->last_iterations > 0;
    if (c->async() && !c->batch) {   // hf_wait_flow(): the host needs m_totalFrameDelta of THIS chain before it decides warp vs copy
        if (!c->ev_flow_done) HF_HIP(c, hipEventCreateWithFlags(&c->ev_flow_done, hipEventDisableTiming));
        HF_HIP(c, hipEventRecord(c->ev_flow_done, s));
        c->flow_done_recorded = true;
    }
    if (c->dual()) {   // tag the flow buffer just written, the tag travels with the buffer through the swap below
        HF_HIP(c, hipEventRecord(c->ev_flow[0], s));
        c->ev_flow_valid[0] = true;
        hipEvent_t te = c->ev_flow[0]; c->ev_flow[0] = c->ev_flow[1]; c->ev_flow[1] = te;
        bool tv = c->ev_flow_valid[0]; c->ev_flow_valid[0] = c->ev_flow_valid[1]; c->ev_flow_valid[1] = tv;
    }
    // opticalFlowCalcSDR.cpp:121-123 : swap so that [1] = newest flow, [0] = previous flow
    int16_t* t = c->blurred[0];
    c->blurred[0] = c->blurred[1];
    c->blurred[1] = t;
    uint32_t* txy = c->blurred_xy[0];
    c->blurred_xy[0] = c->blurred_xy[1];
    c->blurred_xy[1] = txy;
    c->blur_phase ^= 1;
    c->have_flow = true;
    return HF_OK;
}

// Fills the period descriptor of one context: frames N-2 / N-1, the PREVIOUS flow (:154-156), levels, outputs.
// flow_index 1: the period is issued BEFORE the chain of its source period -- the previous flow is still the newest one
void fill_period(hf_ctx* c, int n, const float* t, void* const* outs, hf::WarpPeriod& p, int flow_index) {
    const float scale = c->g.hdr ? 256.0f : 1.0f;
    p.frame12 = c->ring[0]; p.frame21 = c->ring[1];
    p.flow = c->blurred[flow_index]; p.flow_xy = c->blurred_xy[flow_index];
    p.black = c->p.black_level * scale; p.white = c->p.white_level * scale;
    p.n_out = n;
    p.counters = c->counters;
    for (int i = 0; i < n; i++) { p.ts[i] = t[i]; p.outs[i] = outs[i] ? outs[i] : c->out_frame; }
}

int download_common(hf_ctx* c, void* dst, hipMemcpyKind kind) {
    if (int rc = set_device(c)) return rc;
    if (int rc = leave_warp_stream(c)) return rc;
    if (c->out_target != dst) HF_HIP(c, hipMemcpyAsync(dst, c->out_target, c->out_bytes, kind, c->stream));
    if (c->timing()) HF_HIP(c, hipEventRecord(c->ev_warp_end, c->stream));
    if (kind == hipMemcpyDeviceToHost || !c->async()) {
        if (int rc = sync_ctx(c)) return rc;
        float ms = 0.f;
        if (c->warp_started && hipEventElapsedTime(&ms, c->ev_warp_start, c->ev_warp_end) == hipSuccess)
            c->warp_calc_time = (double)ms / 1e3;  // opticalFlowCalcSDR.cpp:36-41
    }
    c->warp_started = false;
    return HF_OK;
}

}  // namespace hfi

extern "C" {

int hf_update_frame(hf_ctx* c, const void* host_frame) {
    HF_CHECK_CTX(c);
    if (!host_frame) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_update_frame: null frame");
    return update_common(c, host_frame, hipMemcpyHostToDevice);
}

int hf_update_frame_device(hf_ctx* c, const void* device_frame) {
    HF_CHECK_CTX(c);
    if (!device_frame) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_update_frame_device: null frame");
    return update_common(c, device_frame, hipMemcpyDeviceToDevice);
}

int hf_update_frame_device_ref(hf_ctx* c, const void* device_frame) {
    HF_CHECK_CTX(c);
    if (!device_frame) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_update_frame_device_ref: null frame");
    return update_common(c, device_frame, hipMemcpyDeviceToDevice, true);
}

int hf_calculate_optical_flow(hf_ctx* c) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    if (int rc = check_flow_params(c)) return rc;
    if (int rc = leave_warp_stream(c)) return rc;
    if (ensure_older_planes(&c, 1, c->stream)) return fail(c, HF_ERR_HIP, "phase-plane launch failed");

    int span = -1;
    if (c->cfg.flags & HF_FLAG_NO_GRAPH) {
        c->tab_mode = choose_tab_mode(&c, 1);
        span = span_begin(c, 2);
        if (int rc = enqueue_flow_chain(c)) return rc;
    } else {
        c->tab_mode = choose_tab_mode(&c, 1);
        const auto key = std::make_tuple(c->ring_phase, c->blur_phase, c->p.search_radius, c->p.delta_scalar, c->p.neighbor_scalar, (int)c->tab_mode);
        auto it = c->graphs.find(key);
        if (it == c->graphs.end()) {
            hipGraph_t graph = nullptr;
            std::shared_lock<std::shared_mutex> capture_lock(g_capture_mutex);
            HF_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue_flow_chain(c);
            const hipError_t e = hipStreamEndCapture(c->stream, &graph);
            capture_lock.unlock();
            if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
            HF_HIP(c, e);
            hipGraphExec_t exec = nullptr;
            HF_HIP(c, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            hipGraphDestroy(graph);
            if (c->graphs.size() >= 96) {  // bound the cache (parameters poked by a settings UI)
                for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
                c->graphs.clear();
            }
            it = c->graphs.emplace(key, exec).first;
        }
        span = span_begin(c, 2);
        HF_HIP(c, hipGraphLaunch(it->second, c->stream));
    }
    span_end(c, span);
    if (int rc = after_flow_enqueued(c, c->stream)) return rc;
    if (!c->async()) return sync_ctx(c);
    return HF_OK;
}

int hf_warp_frames(hf_ctx* c, float t, int mode) {
    HF_CHECK_CTX(c);
    if (t > 1.0f) return fail(c, HF_ERR_INVALID_ARGUMENT, "Error in function warpFrames: blending scalar is greater than 1.0");  // :143-146
    if (mode < 0 || mode > 6) return fail(c, HF_ERR_INVALID_ARGUMENT, "warpFrames: frame output mode %d outside [0, 6]", mode);
    if (int rc = set_device(c)) return rc;
    const float scale = c->g.hdr ? 256.0f : 1.0f;  // opticalFlowCalcHDR.cpp:151-152
    if (!c->warp_started && c->timing()) { HF_HIP(c, hipEventRecord(c->ev_warp_start, c->stream)); c->warp_started = true; }
    if (int rc = enter_warp_stream(c)) return rc;
    if (int rc = guard_output_slot(c, c->out_target, c->warp_stream)) return rc;
    // frames N-2 / N-1 and the PREVIOUS flow (:154-156)
    // profiled launches carry start/stop events of the dispatch itself (hipExtLaunchKernel), i.e. the kernel's
    // execution time as rocprof reports it, not the time the launch spent queued behind other streams
    const int span = span_open(c, 0);
    hf::launch_warp(c->g, c->ring[0], c->ring[1], c->blurred[0], c->blurred_xy[0], c->out_target, t, mode,
                    c->p.black_level * scale, c->p.white_level * scale, c->warp_stream,
                    span >= 0 ? c->spans[span].b : nullptr, span >= 0 ? c->spans[span].e : nullptr);
    if (c->on_warp_stream && !c->in_period) HF_HIP(c, hipEventRecord(c->ev_warps_done, c->warp_stream));
    if (int rc = note_launch(c, c->warp_stream)) return rc;
    HF_HIP(c, hipGetLastError());
    return HF_OK;
}

int hf_copy_frame(hf_ctx* c) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    const float scale = c->g.hdr ? 256.0f : 1.0f;  // opticalFlowCalcHDR.cpp:173-174
    const int idx = c->p.frame_count >= 3 ? 0 : c->p.frame_count >= 2 ? 1 : 2;  // opticalFlowCalcSDR.cpp:173
    if (int rc = leave_warp_stream(c)) return rc;
    if (!c->warp_started && c->timing()) { HF_HIP(c, hipEventRecord(c->ev_warp_start, c->stream)); c->warp_started = true; }
    if (int rc = guard_output_slot(c, c->out_target, c->stream)) return rc;
    const int span = span_begin(c, 1);
    hf::launch_copy(c->g, c->ring[idx], c->out_target, c->p.black_level * scale, c->p.white_level * scale, c->stream);
    span_end(c, span);
    HF_HIP(c, hipGetLastError());
    return note_launch(c, c->stream);
}

int hf_interpolate_period(hf_ctx* c, const void* device_frame, int n_out, const float* t, void* const* device_out, int mode) {
    return hf_interpolate_period_ex(c, device_frame, n_out, t, device_out, mode, 1);
}

int hf_interpolate_period_ex(hf_ctx* c, const void* device_frame, int n_out, const float* t, void* const* device_out, int mode,
                             int update_and_flow) {
    HF_CHECK_CTX(c);
    if (n_out < 0 || (n_out > 0 && (!t || !device_out))) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_interpolate_period: bad argument");
    if (update_and_flow) {
        if (device_frame) if (int rc = hf_update_frame_device_ref(c, device_frame)) return rc;
        if (int rc = hf_calculate_optical_flow(c)) return rc;
    }
    if (int rc = set_device(c)) return rc;
    if (mode < 0 || mode > 6) return fail(c, HF_ERR_INVALID_ARGUMENT, "warpFrames: frame output mode %d outside [0, 6]", mode);
    for (int i = 0; i < n_out; i++)
        if (t[i] > 1.0f) return fail(c, HF_ERR_INVALID_ARGUMENT, "Error in function warpFrames: blending scalar is greater than 1.0");
    // All outputs of the period in one launch when the fast warp kernel applies: the flow is looked up once and
    // the source rows of the later outputs come from L1/L2 instead of HBM (2F + nF bytes instead of n * 3F).
    const bool fuse = n_out >= 2 && !(c->cfg.flags & HF_FLAG_NO_FUSED_WARP);
    int done = 0;
    if (fuse) {
        if (int rc = enter_warp_stream(c)) return rc;
        while (done < n_out) {
            const int n = n_out - done < hf::kMaxWarpOutputs ? n_out - done : hf::kMaxWarpOutputs;
            hf::WarpPeriod p;
            fill_period(c, n, t + done, device_out + done, p);
            for (int i = 0; i < n; i++) if (int rc = guard_output_slot(c, p.outs[i], c->warp_stream)) return rc;
            if (!c->warp_started && c->timing()) { HF_HIP(c, hipEventRecord(c->ev_warp_start, c->stream)); c->warp_started = true; }
            const int span = span_open(c, 0);
            const bool ok = hf::launch_warp_periods(c->g, 1, &p, mode, c->warp_stream,
                                                    span >= 0 ? c->spans[span].b : nullptr, span >= 0 ? c->spans[span].e : nullptr);
            if (!ok) {   // shape not eligible: drop the unused span and fall back to one launch per output
                if (span >= 0) { c->ev_pool.push_back(c->spans[span].b); c->ev_pool.push_back(c->spans[span].e); c->spans.pop_back(); }
                break;
            }
            if (span >= 0) c->spans[span].frames = n;
            HF_HIP(c, hipGetLastError());
            done += n;
        }
        if (done > 0) {
            if (c->on_warp_stream) HF_HIP(c, hipEventRecord(c->ev_warps_done, c->warp_stream));
            if (int rc = note_launch(c, c->warp_stream)) return rc;
        }
    }
    void* const saved = c->out_target;
    c->in_period = true;
    int rc = HF_OK;
    for (int i = done; i < n_out && rc == HF_OK; i++) {
        c->out_target = device_out[i] ? device_out[i] : c->out_frame;
        rc = hf_warp_frames(c, t[i], mode);
    }
    c->in_period = false;
    c->out_target = saved;
    if (rc == HF_OK && done < n_out && c->on_warp_stream) HF_HIP(c, hipEventRecord(c->ev_warps_done, c->warp_stream));
    return rc;
}

int hf_download_frame(hf_ctx* c, void* host_out) {
    HF_CHECK_CTX(c);
    if (!host_out) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_download_frame: null buffer");
    return download_common(c, host_out, hipMemcpyDeviceToHost);
}

int hf_download_frame_device(hf_ctx* c, void* device_out) {
    HF_CHECK_CTX(c);
    if (!device_out) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_download_frame_device: null buffer");
    return download_common(c, device_out, hipMemcpyDeviceToDevice);
}

int hf_set_output_buffer(hf_ctx* c, void* device_out) {
    HF_CHECK_CTX(c);
    c->out_target = device_out ? device_out : c->out_frame;
    return HF_OK;
}

}  // extern "C"
