// hopperrender_amd/csrc/hf_batch.hip -- hf_batch (include/hopperflow.h): up to 32 contexts of identical geometry and parameters issue their
// phase planes, refinement chains and fused period warps as ONE set of launches on one stream (independent frame pairs);
// results per member are those of the single-context calls of hf_calc.hip.  Layout of the ABI: hf_ctx.h.

#include "hf_ctx.h"

using namespace hfi;

static thread_local std::string g_batch_error;

namespace hfi {

int batch_fail(hf_batch* b, int code, const std::string& msg) {
    (b ? b->err : g_batch_error) = "[HopperRender] " + msg;
    return code;
}

// defer: only the grid samples of the new frames now (what the chain of this period reads of them); their full planes are built by
// the next period's warp launch or, failing that, by ensure_older_planes
int batch_update(hf_batch* b, const void* const* device_frames, bool defer) {
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
    if (!device_frames) return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "hf_batch_update_frames_device_ref: null argument");
    hf_ctx* l = b->members[0];
    const int n = (int)b->members.size();
    if (hipSetDevice(l->device) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipSetDevice failed");
    hf::PrepBatch pb{};
    pb.n = n;
    // first pass: everything that can fail, before any member's ring is touched (a failure leaves every member as it was)
    for (int i = 0; i < n; i++) {
        hf_ctx* m = b->members[i];
        if (!device_frames[i]) return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "hf_batch_update_frames_device_ref: null frame");
        if (int rc = leave_warp_stream(m)) return batch_fail(b, rc, m->err);
        if (m->timing() && hipEventRecord(m->ev_upload, b->stream) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipEventRecord failed");
    }
    for (int i = 0; i < n; i++) {
        hf_ctx* m = b->members[i];
        if (m->timing()) m->upload_recorded = true;
        m->ring[0] = const_cast<void*>(device_frames[i]);   // the ring references the caller's frame (hf_update_frame_device_ref)
        pb.frame[i] = m->ring[0];
        pb.pp[i] = m->pp[0];
        m->plane_pending[0] = defer;
    }
    if (defer) hf::launch_prep_grid(l->g, l->pl, pb, b->stream);
    else hf::launch_prep_frames(l->g, l->pl, pb, b->stream);     // the phase planes of all new frames in one launch
    if (hipGetLastError() != hipSuccess) return batch_fail(b, HF_ERR_HIP, "phase-plane launch failed");
    for (hf_ctx* m : b->members) rotate_after_upload(m);
    return HF_OK;
}

// What hf_batch_calculate_optical_flow checks before it enqueues anything: valid flow parameters, equal in all members.
int batch_check_flow_params(hf_batch* b) {
    hf_ctx* l = b->members[0];
    for (hf_ctx* m : b->members) {
        if (int rc = check_flow_params(m)) return batch_fail(b, rc, m->err);
        if (m->p.search_radius != l->p.search_radius || m->p.delta_scalar != l->p.delta_scalar || m->p.neighbor_scalar != l->p.neighbor_scalar)
            return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "hf_batch_calculate_optical_flow: members differ in search radius / delta / neighbor scalar");
    }
    return HF_OK;
}

// before_chain (hf_batch_run_period with deferred phase planes): the period's warps go out AHEAD of the period's chain -- they read
// frames N-2 / N-1 and the previous flow, which the chain does not touch -- and build the full plane of frame N-1 that the chain
// then reads.  Only the one-launch path qualifies; *launched = false means nothing was enqueued and the caller keeps the usual order.
int batch_interpolate(hf_batch* b, const int* n_out, const float* t, void* const* device_out, int mode, bool before_chain, bool* launched) {
    if (launched) *launched = false;
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
    if (!n_out || !t || !device_out) return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "hf_batch_interpolate_period: null argument");
    if (mode < 0 || mode > 6) return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "warpFrames: frame output mode outside [0, 6]");
    hf_ctx* l = b->members[0];
    const int n = (int)b->members.size();
    if (hipSetDevice(l->device) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipSetDevice failed");
    bool one_launch = !l->dual();
    for (int m = 0; m < n; m++) {
        hf_ctx* c = b->members[m];
        if (!c) return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "null context");
        // batch members have no asynchronous host I/O (hf_batch_create / hf_*_async enforce it), so there is no output-ring slot to
        // guard and no side stream to notify here -- the one-launch path relies on that
        if (c->io_in) return batch_fail(b, HF_ERR_STATE, "hf_batch_interpolate_period: a member uses asynchronous host I/O");
        one_launch = one_launch && !(c->cfg.flags & HF_FLAG_NO_FUSED_WARP);
        if (n_out[m] < 0 || n_out[m] > HF_MAX_PERIOD_OUTPUTS) return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "hf_batch_interpolate_period: n_out outside [0, 6]");
        for (int i = 0; i < n_out[m]; i++)
            if (t[m * HF_MAX_PERIOD_OUTPUTS + i] > 1.0f)
                return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "Error in function warpFrames: blending scalar is greater than 1.0");
        one_launch = one_launch && n_out[m] >= 1;
    }
    if (one_launch) {
        // every member's period in ONE launch on the batch stream (single-stream members: program order does the rest)
        hf::WarpPeriod periods[hf::kMaxFlowBatch];
        for (int m = 0; m < n; m++) {
            hf_ctx* c = b->members[m];
            fill_period(c, n_out[m], t + m * HF_MAX_PERIOD_OUTPUTS, device_out + m * HF_MAX_PERIOD_OUTPUTS, periods[m], before_chain ? 1 : 0);
            if (before_chain && c->plane_pending[1]) periods[m].plane21 = c->pp[1];
            if (!c->warp_started && c->timing()) {   // m_warpCalcTime span of the member (opticalFlowCalcSDR.cpp:36-41), as in hf_warp_frames
                if (hipEventRecord(c->ev_warp_start, b->stream) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipEventRecord failed");
                c->warp_started = true;
            }
        }
        const int span = hf::t_launch_observer == &b->tl ? -1 : span_open(l, 0);   // (an observed launch carries the timeline's events, not a profile span's)
        bool built[hf::kMaxFlowBatch];
        if (hf::launch_warp_periods(l->g, n, periods, mode, b->stream, span >= 0 ? l->spans[span].b : nullptr, span >= 0 ? l->spans[span].e : nullptr,
                                    before_chain ? &l->pl : nullptr, built)) {
            if (span >= 0) { int f = 0; for (int m = 0; m < n; m++) f += n_out[m]; l->spans[span].frames = f; }
            if (launched) *launched = true;   // from here on the period's warps are enqueued: an error is final, never a reason to issue them again
            if (hipGetLastError() != hipSuccess) return batch_fail(b, HF_ERR_HIP, "fused warp launch failed");
            for (int m = 0; m < n; m++) if (built[m]) b->members[m]->plane_pending[1] = false;
            return HF_OK;
        }
        if (span >= 0) { l->ev_pool.push_back(l->spans[span].b); l->ev_pool.push_back(l->spans[span].e); l->spans.pop_back(); }
    }
    if (before_chain) return HF_OK;   // not eligible for one launch: the caller issues the period after the chain, as usual
    for (int m = 0; m < n; m++)   // not eligible (diagnostic modes, odd shapes, dual-stream members): member by member
        if (int rc = hf_interpolate_period_ex(b->members[m], nullptr, n_out[m], t + m * HF_MAX_PERIOD_OUTPUTS, device_out + m * HF_MAX_PERIOD_OUTPUTS, mode, 0))
            return batch_fail(b, rc, b->members[m]->err);
    return HF_OK;
}

}  // namespace hfi

extern "C" {

const char* hf_batch_last_error(const hf_batch* b) { return b ? b->err.c_str() : g_batch_error.c_str(); }

int hf_batch_create(hf_ctx* const* members, int n, hf_batch** out) {
    if (!members || !out || n < 1) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_batch_create: bad argument");
    *out = nullptr;
    if (n > hf::kMaxFlowBatch) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_batch_create: at most " + std::to_string(hf::kMaxFlowBatch) + " members");
    hf_ctx* l = members[0];
    for (int i = 0; i < n; i++) {
        hf_ctx* m = members[i];
        if (!m) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_batch_create: null member");
        for (int j = 0; j < i; j++) if (members[j] == m) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_batch_create: duplicate member");
        if (m->batch) return batch_fail(nullptr, HF_ERR_STATE, "hf_batch_create: member " + std::to_string(i) + " already belongs to a batch");
        const hf::Geom &a = l->g, &b = m->g;
        const bool same = a.hdr == b.hdr && a.H == b.H && a.W == b.W && a.in_stride == b.in_stride && a.out_stride == b.out_stride &&
                          a.rs == b.rs && m->device == l->device && m->cfg.iterations == l->cfg.iterations &&
                          m->cfg.blur_radius == l->cfg.blur_radius && !m->sadtab == !l->sadtab;
        if (!same) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_batch_create: members differ in geometry, device, iterations, blur radius or HF_FLAG_NO_SAD_REUSE");
        if (!m->async() || m->io_in || m->dual() != l->dual())
            return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_batch_create: members must be HF_FLAG_ASYNC contexts (all single-stream or all HF_FLAG_DUAL_STREAM) without async host I/O");
    }
    if (hipSetDevice(l->device) != hipSuccess) return batch_fail(nullptr, HF_ERR_HIP, "hf_batch_create: hipSetDevice failed");
    for (int i = 0; i < n; i++)   // before any member is touched: a failure leaves every context as it was
        if (int rc = sync_ctx(members[i])) return batch_fail(nullptr, rc, "hf_batch_create: member sync failed: " + members[i]->err);
    hf_batch* b = new (std::nothrow) hf_batch();
    if (!b) return batch_fail(nullptr, HF_ERR_OUT_OF_MEMORY, "hf_batch_create: host allocation failed");
    // A stream of the batch's own, of the HIGHEST priority.  Not for the priority: the runtime keeps one pool of hardware queues
    // per priority level and hands a new stream the queue of its pool with the fewest users.  The members' streams (and
    // everybody else's) are normal-priority ones, so the batch streams of a process are alone in their pool and the first
    // GPU_MAX_HW_QUEUES of them sit on different hardware queues whatever was created before.  (With the leader's stream, two
    // batches whose leaders were 32 streams apart shared ONE queue and ran strictly one after the other: 64 x 32 at 103 k
    // instead of 115 k frames/s; a normal-priority stream of the batch's own did the same at 48 x 24.)
    // Side effect (include/hopperflow.h): the priority is real -- batch work is scheduled ahead of the normal-priority streams
    // of the process.  HF_FLAG_BATCH_NORMAL_PRIORITY on the leader opts out (and gives up the private queue pool).
    {
        int prio_low = 0, prio_high = 0;
        const bool want_high = !(l->cfg.flags & HF_FLAG_BATCH_NORMAL_PRIORITY) && hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) == hipSuccess;
        if (!want_high || hipStreamCreateWithPriority(&b->stream, hipStreamNonBlocking, prio_high) != hipSuccess) {
            (void)hipGetLastError();
            b->stream = nullptr;
            if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
                delete b;
                return batch_fail(nullptr, HF_ERR_HIP, "hf_batch_create: cannot create the batch stream");
            }
        }
    }
    if (l->dual()) {
        // the members' warps go to a few shared streams (round robin) instead of one stream per member: the device
        // runs only a handful of hardware queues side by side (DESIGN.md "Hardware queues")
        const int nws = n < 3 ? n : 3;
        for (int i = 0; i < nws; i++) {
            hipStream_t ws = nullptr;
            if (hipStreamCreateWithFlags(&ws, hipStreamNonBlocking) != hipSuccess) {
                for (hipStream_t x : b->warp_streams) hipStreamDestroy(x);
                hipStreamDestroy(b->stream);
                delete b;
                return batch_fail(nullptr, HF_ERR_HIP, "hf_batch_create: cannot create a warp stream");
            }
            b->warp_streams.push_back(ws);
        }
    }
    for (int i = 0; i < n; i++) {
        hf_ctx* m = members[i];
        b->members.push_back(m);
        b->own_streams.push_back(m->stream);
        b->own_warp_streams.push_back(m->warp_stream);
        // one stream for the whole batch: the members' prep / warp launches and the batched chain stay in program order
        for (auto& kv : m->graphs) hipGraphExecDestroy(kv.second);   // captured on the member's own stream
        m->graphs.clear();
        m->stream = b->stream;
        m->warp_stream = m->dual() ? b->warp_streams[(size_t)i % b->warp_streams.size()] : b->stream;
        m->batch = b;
    }
    // Deferred phase planes: where the batched period warp is the workgroup-staged kernel it can build the full plane of the frame
    // it reads anyway (plane-building workgroups of warp_wg_kernel, hf_kernels.hip); hf_batch_run_period then only samples the grid
    // at update time.
    b->defer_planes = !l->dual() && !(l->cfg.flags & HF_FLAG_BATCH_EAGER_PLANES) && hf::warp_period_can_build_planes(l->g, l->pl, n);
    for (int i = 0; i < n; i++) b->defer_planes = b->defer_planes && !(members[i]->cfg.flags & HF_FLAG_NO_FUSED_WARP);
    *out = b;
    return HF_OK;
}

void hf_batch_destroy(hf_batch* b) {
    if (!b) return;
    if (!b->members.empty()) hipSetDevice(b->members[0]->device);
    for (hf_ctx* m : b->members) leave_warp_stream(m);   // the batch stream waits for every member's last warps
    if (b->stream) hipStreamSynchronize(b->stream);
    for (hipStream_t ws : b->warp_streams) hipStreamSynchronize(ws);
    for (auto& kv : b->graphs) hipGraphExecDestroy(kv.second);
    for (size_t i = 0; i < b->members.size(); i++) {
        hf_ctx* m = b->members[i];
        for (auto& kv : m->graphs) hipGraphExecDestroy(kv.second);
        m->graphs.clear();
        m->stream = b->own_streams[i];
        m->warp_stream = b->own_warp_streams[i];
        m->on_warp_stream = false;
        m->batch = nullptr;
    }
    for (hipStream_t ws : b->warp_streams) hipStreamDestroy(ws);
    if (b->stream) hipStreamDestroy(b->stream);
    for (hipEvent_t e : b->tl.events) hipEventDestroy(e);
    delete b;
}

int hf_batch_update_frames_device_ref(hf_batch* b, const void* const* device_frames) { return batch_update(b, device_frames, false); }

int hf_batch_calculate_optical_flow(hf_batch* b) {
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
    hf_ctx* l = b->members[0];
    const int n = (int)b->members.size();
    if (hipSetDevice(l->device) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipSetDevice failed");
    if (int rc = batch_check_flow_params(b)) return rc;
    l->tab_mode = choose_tab_mode(b->members.data(), n);
    std::vector<int> key = {l->p.search_radius, l->p.delta_scalar, l->p.neighbor_scalar, (int)l->tab_mode};
    for (hf_ctx* m : b->members) {
        if (int rc = leave_warp_stream(m)) return batch_fail(b, rc, m->err);
        key.push_back(m->ring_phase * 2 + m->blur_phase);
    }
    if (ensure_older_planes(b->members.data(), n, b->stream)) return batch_fail(b, HF_ERR_HIP, "phase-plane launch failed");
    if (hf::t_launch_observer == &b->tl) {
        // timeline: the chain's launches one by one, each with the events of its own dispatch (a graph replay has no per-node timestamps)
        if (int rc = enqueue_flow_chain(b->members.data(), n, b->stream)) return batch_fail(b, rc, l->err);
        for (hf_ctx* m : b->members)
            if (int rc = after_flow_enqueued(m, b->stream)) return batch_fail(b, rc, m->err);
        return HF_OK;
    }
    auto it = b->graphs.find(key);
    if (it == b->graphs.end()) {
        hipGraph_t graph = nullptr;
        std::shared_lock<std::shared_mutex> capture_lock(g_capture_mutex);
        if (hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipStreamBeginCapture failed");
        const int rc = enqueue_flow_chain(b->members.data(), n, b->stream);
        const hipError_t e = hipStreamEndCapture(b->stream, &graph);
        capture_lock.unlock();
        if (rc || e != hipSuccess) { if (graph) hipGraphDestroy(graph); return batch_fail(b, rc ? rc : HF_ERR_HIP, rc ? l->err : "hipStreamEndCapture failed"); }
        hipGraphExec_t exec = nullptr;
        const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ei != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipGraphInstantiate failed");
        if (b->graphs.size() >= 96) {
            for (auto& kv : b->graphs) hipGraphExecDestroy(kv.second);
            b->graphs.clear();
        }
        it = b->graphs.emplace(key, exec).first;
    }
    const int span = span_begin(l, 2);   // the leader's profile carries the batch (one span = n chains)
    if (hipGraphLaunch(it->second, b->stream) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipGraphLaunch failed");
    span_end(l, span);
    if (span >= 0) l->spans[span].frames = n;
    for (hf_ctx* m : b->members)
        if (int rc = after_flow_enqueued(m, b->stream)) return batch_fail(b, rc, m->err);
    return HF_OK;
}

int hf_batch_size(const hf_batch* b) { return b ? (int)b->members.size() : 0; }

int hf_batch_interpolate_period(hf_batch* b, const int* n_out, const float* t, void* const* device_out, int mode) {
    return batch_interpolate(b, n_out, t, device_out, mode, false, nullptr);
}

int hf_batch_run_period(hf_batch* b, const void* const* device_frames, int calculate_flow, const int* n_out, const float* t,
                        void* const* device_out, int mode) {
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
    struct ObserverGuard {   // timeline on: every launch of this call carries its own start / stop events (hf_kernels.h HF_LAUNCH)
        hf_batch* b;
        explicit ObserverGuard(hf_batch* x) : b(nullptr) {
            if (!x->tl.active) return;
            if (x->tl.skip > 0) { x->tl.skip--; return; }      // armed, not recording yet
            b = x;
            hf::t_launch_observer = &b->tl;
        }
        ~ObserverGuard() {
            if (!b) return;
            hf::t_launch_observer = nullptr;
            b->tl.period++;
            if (b->tl.recs.size() + 32 > b->tl.capacity) b->tl.active = false;   // no room for another whole period: back to graph replays
        }
    } observer_guard(b);
    if (device_frames) if (int rc = batch_update(b, device_frames, b->defer_planes)) return rc;
    // Deferred phase planes: a period whose older frame still lacks its full plane issues its warps FIRST (they do not depend on
    // this period's chain) and lets that launch build the plane; same results as the order of the three calls.
    bool warped = false;
    if (n_out && calculate_flow && b->defer_planes && mode >= 0 && mode <= 2) {
        // The early warps must not write the caller's output buffers in a period whose flow calculation is going to be refused: the three
        // separate calls would have stopped at the chain, before any warp.  So the chain's own argument checks come first.
        if (int rc = batch_check_flow_params(b)) return rc;
        bool pending = false;
        for (hf_ctx* m : b->members) pending = pending || m->plane_pending[1];
        // (an argument error of this early attempt -- nothing enqueued, `warped` false -- is not reported here: the period then takes the usual
        //  order below, which reports the same error where the three separate calls would, after the update and the chain; a launch that was
        //  enqueued and failed is reported at once)
        if (pending) if (int rc = batch_interpolate(b, n_out, t, device_out, mode, true, &warped)) { if (warped) return rc; }
    }
    if (calculate_flow) if (int rc = hf_batch_calculate_optical_flow(b)) return rc;
    if (n_out && !warped) if (int rc = hf_batch_interpolate_period(b, n_out, t, device_out, mode)) return rc;
    return HF_OK;
}

int hf_batch_defers_planes(const hf_batch* b) { return b && b->defer_planes ? 1 : 0; }

// ---- timeline: start / stop of every dispatch of a batch on the device's clock, no profiler attached ----
namespace {
std::mutex g_tl_mutex;
std::map<int, hipEvent_t> g_tl_reference;   // per device: the zero of every batch's timeline in this process
}
int hf_batch_timeline_enable(hf_batch* b, int max_launches, int skip_periods) {
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
    hf_ctx* l = b->members[0];
    if (hipSetDevice(l->device) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipSetDevice failed");
    if (max_launches < 0 || max_launches > (1 << 20) || skip_periods < 0 || (max_launches > 0 && max_launches < 32))
        return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "hf_batch_timeline_enable: max_launches must be 0 or in [32, 2^20] (a period needs up to 32 free records), skip_periods >= 0");
    if (hipStreamSynchronize(b->stream) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipStreamSynchronize failed");
    b->tl.active = false;
    b->tl.recs.clear();
    b->tl.period = 0;
    b->tl.dropped = 0;
    b->tl.capacity = 0;
    if (max_launches == 0) {
        for (hipEvent_t e : b->tl.events) hipEventDestroy(e);
        b->tl.events.clear();
        return HF_OK;
    }
    {
        std::lock_guard<std::mutex> lock(g_tl_mutex);
        if (!g_tl_reference.count(l->device)) {
            hipEvent_t ref = nullptr;
            if (hipEventCreate(&ref) != hipSuccess || hipEventRecord(ref, b->stream) != hipSuccess || hipEventSynchronize(ref) != hipSuccess)
                return batch_fail(b, HF_ERR_HIP, "hf_batch_timeline_enable: cannot record the reference event");
            g_tl_reference[l->device] = ref;
        }
    }
    while (b->tl.events.size() < 2 * (size_t)max_launches) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return batch_fail(b, HF_ERR_OUT_OF_MEMORY, "hf_batch_timeline_enable: hipEventCreate failed");
        b->tl.events.push_back(e);
    }
    b->tl.recs.reserve((size_t)max_launches);
    b->tl.capacity = (size_t)max_launches;
    b->tl.skip = skip_periods;
    b->tl.active = true;
    return HF_OK;
}

int hf_batch_timeline_read(hf_batch* b, hf_timeline_record* out, int capacity, int* n_records) {
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
    if (!n_records || (capacity > 0 && !out)) return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "hf_batch_timeline_read: null argument");
    hf_ctx* l = b->members[0];
    if (hipSetDevice(l->device) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipSetDevice failed");
    if (hipStreamSynchronize(b->stream) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipStreamSynchronize failed");
    for (hipStream_t ws : b->warp_streams)      // HF_FLAG_DUAL_STREAM members: the per-member warps of an observed period run there
        if (hipStreamSynchronize(ws) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipStreamSynchronize (warp stream) failed");
    hipEvent_t ref = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_tl_mutex);
        auto it = g_tl_reference.find(l->device);
        if (it != g_tl_reference.end()) ref = it->second;
    }
    *n_records = (int)b->tl.recs.size();
    if (!ref) return b->tl.recs.empty() ? HF_OK : batch_fail(b, HF_ERR_STATE, "hf_batch_timeline_read: no reference event");
    const int n = *n_records < capacity ? *n_records : capacity;
    for (int i = 0; i < n; i++) {
        const hf_timeline::Rec& r = b->tl.recs[(size_t)i];
        float t0 = 0.f, t1 = 0.f, d = 0.f;
        hf_timeline_record& o = out[i];
        std::memset(&o, 0, sizeof(o));
        std::strncpy(o.kernel, r.name, sizeof(o.kernel) - 1);
        o.period = r.period;
        if (hipEventElapsedTime(&t0, ref, r.b) != hipSuccess || hipEventElapsedTime(&t1, ref, r.e) != hipSuccess ||
            hipEventElapsedTime(&d, r.b, r.e) != hipSuccess) {
            (void)hipGetLastError();      // events of a launch that failed or never ran: flag the record, keep the others
            o.flags = 1;
            continue;
        }
        o.start_ms = (double)t0;
        o.end_ms = (double)t1;
        o.duration_ms = (double)d;
    }
    return HF_OK;
}

uint64_t hf_batch_timeline_dropped(const hf_batch* b) { return b ? b->tl.dropped : 0; }

int hf_batch_sync(hf_batch* b) {
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
    for (hf_ctx* m : b->members)
        if (int rc = hf_sync(m)) return batch_fail(b, rc, m->err);
    return HF_OK;
}

}  // extern "C"
