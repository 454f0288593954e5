// hopperrender_amd/csrc/hf_capi.hip -- the C ABI of include/hopperflow.h on top of the gfx950
// kernels: device memory, the 3-frame ring, the 16-step refinement chain as a cached hipGraph,
// HIP-event timing with the reference's span definitions.
//
// Host orchestration restated from the reference's opticalFlowCalcSDR.cpp / opticalFlowCalcHDR.cpp
// (cited per function); the architecture differs on purpose:
//   * every uploaded frame is re-laid out once as mirror-padded phase planes, so the candidates of a run
//     of grid pixels are consecutive bytes (hf_flow.hip);
//   * offsets are kept per window in one small table per level, not per pixel;
//   * ONE launch per level (X and Y step fused) for windows <= 32, two per axis for larger windows
//     (reference: fill + calcDeltaSums + determineLowestLayer + adjustOffsetArray = 4 enqueues per step,
//     2.6-8.3 MB of fills);
//   * m_totalFrameDelta is produced on the device and copied to pinned memory inside the graph
//     (reference: blocking 4-byte readback in the middle of the chain, opticalFlowCalcSDR.cpp:91-94);
//   * the whole chain replays as one hipGraph keyed by (ring phase, search radius, scalars).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/config.h"
#include "../../include/hopperflow.h"
#include "hf_kernels.h"

namespace {

thread_local std::string g_create_error;

constexpr int kMinSearchRadius = MIN_SEARCH_RADIUS;    // include/config.h (reference config.h:8)
constexpr int kMaxSearchRadius = MAX_SEARCH_RADIUS;    // config.h:9
constexpr int kCalcTimeInterval = CALC_TIME_INTERVAL;  // config.h:17
static_assert(kMaxSearchRadius <= 16, "the chain kernels keep 16 candidates per step in registers");
constexpr int kMaxSteps = 32;          // 2 * log2(max window)

std::shared_mutex g_capture_mutex;   // shared: a stream capture is in progress; exclusive: a legacy-stream copy (util_copy)

}  // namespace

struct hf_ctx {
    hf::Geom g{};
    hf_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;                      // stream the context issues on (a batch's shared stream while it is a member)
    hipStream_t own_stream = nullptr;                  // the stream this context created and destroys
    hipStream_t warp_stream = nullptr;                 // == stream unless HF_FLAG_DUAL_STREAM
    hipStream_t own_warp_stream = nullptr;             // HF_FLAG_DUAL_STREAM: this context's second stream
    hipEvent_t ev_chain_done = nullptr, ev_warps_done = nullptr;
    struct hf_batch* batch = nullptr;                  // the batch this context is a member of
    hipEvent_t ev_flow[2] = {nullptr, nullptr};        // recorded behind the chain that wrote blurred[i] (swapped with it)
    bool ev_flow_valid[2] = {false, false};
    bool dual() const { return (cfg.flags & HF_FLAG_DUAL_STREAM) != 0; }
    bool on_warp_stream = false;                       // warp stream currently ordered after `stream`
    bool in_period = false;                            // inside hf_interpolate_period (one completion event for all its warps)
    std::string err;

    // public fields of the reference object (opticalFlowCalc.h:27-48)
    hf_params p{};
    uint32_t total_frame_delta = 0;
    double ofc_calc_time = 0, ofc_avg = 0, ofc_peak = 0, ofc_sum = 0, warp_calc_time = 0;
    int ofc_count = 0;

    // device memory (reference buffers: opticalFlowCalcSDR.cpp:272-280)
    size_t in_bytes = 0, out_bytes = 0, plane_elems = 0;
    void* ring[3] = {nullptr, nullptr, nullptr};       // m_inputFrameArray, ring[2] = newest (may point at caller memory)
    void* ring_store[3] = {nullptr, nullptr, nullptr}; // the context's own frame buffers, rotating with the ring
    uint32_t* pp[3] = {nullptr, nullptr, nullptr};     // phase plane of each ring frame (hf_flow.hip)
    bool plane_pending[3] = {false, false, false};     // deferred build (hf_batch_run_period): pp[i] holds only the grid samples so far
    hf::PhaseLayout pl{};
    void* out_frame = nullptr;                         // m_outputFrameArray
    void* out_target = nullptr;                        // where warp/copy write (out_frame or caller's)
    int16_t* tables = nullptr;                         // per-level window offsets (replaces the per-pixel m_offsetArray)
    size_t tables_bytes = 0;
    std::vector<hf::FlowLevel> levels;                 // level k = window size initial_window >> k
    hf::FlowLevel last_level{};                        // level the last chain ended on (tx == nullptr: no step ran)
    int16_t* off_view = nullptr;                       // scratch [2][lh][lw] for hf_read_offsets
    int16_t* blurred[2] = {nullptr, nullptr};          // m_blurredOffsetArray
    uint32_t* blurred_xy[2] = {nullptr, nullptr};      // the same flow packed x | y << 16 (fast warp path)
    uint32_t* sums = nullptr;                          // [kMaxSteps][n_windows_max][16]
    size_t sums_bytes = 0;
    size_t sums_stride = 0;                            // elements per step
    uint32_t* d_total_delta = nullptr;                 // device view of h_total_delta (mapped pinned memory)
    uint32_t* h_total_delta = nullptr;                 // pinned host slot the chain writes m_totalFrameDelta into
    float* d_probe = nullptr;

    // asynchronous host I/O (hf_update_frame_async / hf_download_frame_async): side streams, created lazily
    static constexpr int kOutRing = 3;
    hipStream_t io_in = nullptr, io_out = nullptr;
    hipEvent_t ev_h2d = nullptr, ev_last_launch = nullptr, ev_out_ready = nullptr;
    hipEvent_t ev_slot_prep[3] = {nullptr, nullptr, nullptr};   // prep of ring_store[i] finished (rotates with the ring)
    hipEvent_t ev_d2h[kOutRing] = {nullptr, nullptr, nullptr};
    bool d2h_pending[kOutRing] = {false, false, false};
    void* out_ring[kOutRing] = {nullptr, nullptr, nullptr};      // [0] == out_frame
    int out_idx = 0;
    bool have_last_launch = false;
    // completion of the asynchronous readbacks, for streaming hosts (hf_wait_download): one event per download, ring of kDlRing
    static constexpr int kDlRing = 64;
    hipEvent_t ev_dl[kDlRing] = {};
    uint64_t dl_issued = 0;
    hipEvent_t ev_flow_done = nullptr;                 // behind the last chain of an asynchronous context (hf_wait_flow)
    bool flow_done_recorded = false;

    int ring_phase = 0;   // number of rotations mod 3 (graph key)
    int blur_phase = 0;   // number of swaps mod 2
    bool have_flow = false;
    bool delta_pending = false;
    int last_iterations = 0, initial_window = 0;

    // timing (reference spans: opticalFlowCalcSDR.cpp:36-41,119-127)
    hipEvent_t ev_upload = nullptr, ev_flow_end = nullptr, ev_warp_start = nullptr, ev_warp_end = nullptr;
    hipEvent_t ev_user0 = nullptr, ev_user1 = nullptr;
    bool upload_recorded = false, flow_timing_pending = false, warp_started = false;

    std::map<std::tuple<int, int, int, int, int>, hipGraphExec_t> graphs;

    // HF_FLAG_PROFILE: event pairs around warp / copy / flow-chain launches
    struct Span { hipEvent_t b, e; int kind; hipStream_t stream; int frames = 1; };
    std::vector<hipEvent_t> ev_pool;
    std::vector<Span> spans;
    hf_profile prof{};
    int prof_every[3] = {1, 1, 1};   // sampling interval per span kind (warp, copy, flow chain)
    unsigned prof_seen[3] = {0, 0, 0};
    bool profiling() const { return (cfg.flags & HF_FLAG_PROFILE) != 0; }

    bool async() const { return (cfg.flags & HF_FLAG_ASYNC) != 0; }
    bool timing() const { return (cfg.flags & HF_FLAG_NO_TIMING) == 0; }   // record the reference's timing events
};

namespace {

int fail(hf_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    std::string msg = std::string("[HopperRender] ") + buf;  // same prefix as the reference's exceptions
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HF_HIP(c, call)                                                                              \
    do {                                                                                             \
        hipError_t _e = (call);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return fail((c), _e == hipErrorOutOfMemory ? HF_ERR_OUT_OF_MEMORY : HF_ERR_HIP,          \
                        "HIP error %d (%s) in %s at %s:%d", (int)_e, hipGetErrorString(_e), #call,  \
                        __FILE__, __LINE__);                                                         \
    } while (0)

#define HF_CHECK_CTX(c) \
    if (!(c)) return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null context")

int set_device(hf_ctx* c) {
    HF_HIP(c, hipSetDevice(c->device));
    return HF_OK;
}

// opticalFlowCalcSDR.cpp:49-59
int initial_window(int lw, int lh) {
    int max_dim = lw > lh ? lw : lh;
    int ws;
    if (max_dim && !(max_dim & (max_dim - 1))) {
        ws = max_dim;
    } else {
        while (max_dim & (max_dim - 1)) max_dim &= (max_dim - 1);
        ws = max_dim << 1;
    }
    return ws / 2;
}

int ilog2(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) l++;
    return l;
}

int effective_iterations(const hf_ctx* c) {
    int iters = ilog2(initial_window(c->g.lw, c->g.lh));             // opticalFlowCalcSDR.cpp:62-65
    if (c->cfg.iterations > 0 && c->cfg.iterations < iters) iters = c->cfg.iterations;
    return iters;
}

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
    }
    hf::launch_blur_flow(g, bb, c->cfg.blur_radius, (int)(c->sums_bytes / sizeof(uint32_t)), s);  // :115-116
    HF_HIP(c, hipGetLastError());
    return HF_OK;
}
int enqueue_flow_chain(hf_ctx* c) { return enqueue_flow_chain(&c, 1, c->stream); }

// Deferred phase planes (hf_batch_run_period): the chain reads the FULL plane of frame N-1.  Members whose pp[1] still holds only
// the grid samples -- no warp launch took the build up -- get it from the stand-alone plane kernel now, in one launch.
static int ensure_older_planes(hf_ctx* const* cs, int n, hipStream_t s) {
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

hipEvent_t pool_event(hf_ctx* c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// Opens a profiled span on the stream; returns the index of the span or -1.
int span_begin(hf_ctx* c, int kind, hipStream_t stream = nullptr) {
    if (!c->profiling()) return -1;
    if ((c->prof_seen[kind]++ % (unsigned)c->prof_every[kind]) != 0) return -1;
    hf_ctx::Span s{pool_event(c), pool_event(c), kind, nullptr, 1};
    if (!s.b || !s.e) return -1;
    s.stream = stream ? stream : c->stream;
    hipEventRecord(s.b, s.stream);
    c->spans.push_back(s);
    return (int)c->spans.size() - 1;
}
// Span whose two events are filled in by the launch itself (hipExtLaunchKernelGGL start/stop events).
int span_open(hf_ctx* c, int kind) {
    if (!c->profiling()) return -1;
    if ((c->prof_seen[kind]++ % (unsigned)c->prof_every[kind]) != 0) return -1;
    hf_ctx::Span s{pool_event(c), pool_event(c), kind, nullptr, 1};
    if (!s.b || !s.e) return -1;
    c->spans.push_back(s);
    return (int)c->spans.size() - 1;
}
void span_end(hf_ctx* c, int idx) {
    if (idx >= 0) hipEventRecord(c->spans[idx].e, c->spans[idx].stream);
}
void collect_spans(hf_ctx* c) {  // stream must be idle
    for (auto& s : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.b, s.e) == hipSuccess) {
            if (s.kind == 0) { c->prof.warp_launches++; c->prof.warp_ms += ms; c->prof.warp_frames += (uint64_t)s.frames; }
            else if (s.kind == 1) { c->prof.copy_launches++; c->prof.copy_ms += ms; }
            else { c->prof.flow_chains += (uint64_t)s.frames; c->prof.flow_ms += ms; }   // a batch span covers n chains
        }
        c->ev_pool.push_back(s.b);
        c->ev_pool.push_back(s.e);
    }
    c->spans.clear();
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

int sync_ctx(hf_ctx* c) {
    if (int rc = leave_warp_stream(c)) return rc;
    HF_HIP(c, hipStreamSynchronize(c->stream));
    if (c->io_in) { HF_HIP(c, hipStreamSynchronize(c->io_in)); HF_HIP(c, hipStreamSynchronize(c->io_out)); }
    collect_spans(c);
    if (c->delta_pending) { c->total_frame_delta = *c->h_total_delta; c->delta_pending = false; }
    finish_flow_timing(c);
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

int io_init(hf_ctx* c) {
    if (c->io_in) return HF_OK;
    HF_HIP(c, hipStreamCreateWithFlags(&c->io_in, hipStreamNonBlocking));
    HF_HIP(c, hipStreamCreateWithFlags(&c->io_out, hipStreamNonBlocking));
    HF_HIP(c, hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming));
    HF_HIP(c, hipEventCreateWithFlags(&c->ev_last_launch, hipEventDisableTiming));
    HF_HIP(c, hipEventCreateWithFlags(&c->ev_out_ready, hipEventDisableTiming));
    for (auto& e : c->ev_slot_prep) HF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev_d2h) HF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev_dl) HF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->out_ring[0] = c->out_frame;
    for (int i = 1; i < hf_ctx::kOutRing; i++) HF_HIP(c, hipMalloc(&c->out_ring[i], c->out_bytes));
    return HF_OK;
}

// Before a warp/copy writes into an output-ring slot: wait for the asynchronous readback that still uses it.
int guard_output_slot(hf_ctx* c, const void* target, hipStream_t launch_stream) {
    if (!c->io_out) return HF_OK;
    for (int i = 0; i < hf_ctx::kOutRing; i++)
        if (target == c->out_ring[i] && c->d2h_pending[i]) {
            HF_HIP(c, hipStreamWaitEvent(launch_stream, c->ev_d2h[i], 0));
            c->d2h_pending[i] = false;
        }
    return HF_OK;
}
int note_launch(hf_ctx* c, hipStream_t launch_stream) {   // remembers "the frames/flow of the ring are being read up to here"
    if (!c->io_in) return HF_OK;
    HF_HIP(c, hipEventRecord(c->ev_last_launch, launch_stream));
    c->have_last_launch = true;
    return HF_OK;
}

// by_reference: the ring slot points at the caller's device frame instead of receiving a copy
int update_common(hf_ctx* c, const void* src, hipMemcpyKind kind, bool by_reference = false) {
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

}  // namespace

extern "C" {

int hf_abi_version(void) { return HF_ABI_VERSION; }

const char* hf_last_error(const hf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// detectDevices' selection rule on a capability table (opticalFlowCalc.cpp:67-93): the first entry with enough memory, >= 2 KB
// of LDS per workgroup and 16 x 16 workgroups -- plus, for this build's kernels, 64-wide wavefronts.
int hf_select_device(const hf_device_caps* caps, int n, uint64_t required_vram_bytes, char* why_not, size_t why_not_size) {
    if (why_not && why_not_size) why_not[0] = 0;
    if (!caps || n < 1) return -1;
    for (int d = 0; d < n; d++)
        if (caps[d].vram_bytes >= required_vram_bytes && caps[d].lds_bytes_per_workgroup >= 2048 && caps[d].max_threads_per_workgroup >= 256 &&
            caps[d].wavefront_size == 64)
            return d;
    if (why_not && why_not_size) {   // the reference reports the LAST device it looked at (:98-108)
        const hf_device_caps& k = caps[n - 1];
        size_t o = 0;
        auto add = [&](const char* fmt, auto... v) { if (o < why_not_size) { const int w = snprintf(why_not + o, why_not_size - o, fmt, v...); if (w > 0) o += (size_t)w; } };
        if (k.vram_bytes < required_vram_bytes)
            add("Not enough VRAM available! Required: %llu MB, Available: %llu MB. ", (unsigned long long)(required_vram_bytes / 1024 / 1024), (unsigned long long)(k.vram_bytes / 1024 / 1024));
        if (k.lds_bytes_per_workgroup < 2048)
            add("Not enough shared memory available! Required: 2048 bytes, Available: %llu bytes. ", (unsigned long long)k.lds_bytes_per_workgroup);
        if (k.max_threads_per_workgroup < 256)
            add("Not enough work group sizes available! Required: 16, 16, 1 (256 threads), Available: %d threads. ", k.max_threads_per_workgroup);
        if (k.wavefront_size != 64) add("Wavefront size %d, the kernels need 64. ", k.wavefront_size);
    }
    return -1;
}

int hf_get_device(const hf_ctx* c) { return c ? c->device : -1; }

int hf_create(const hf_config* cfg, hf_ctx** out_ctx) {
    if (!cfg || !out_ctx) return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_create: null argument");
    *out_ctx = nullptr;
    if (cfg->struct_size != sizeof(hf_config))
        return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_create: hf_config.struct_size %u != %zu", cfg->struct_size, sizeof(hf_config));
    if (cfg->frame_height < 4 || cfg->frame_width < 4 || (cfg->frame_height & 1) || (cfg->frame_width & 1))
        return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_create: frame size %dx%d must be even and >= 4", cfg->frame_width, cfg->frame_height);
    if (cfg->max_calc_res < 1) return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_create: max_calc_res must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, HF_ERR_NO_DEVICE, "Error in function detectDevices: no HIP device available");
    if (cfg->device_index < -1 || cfg->device_index >= ndev)
        return fail(nullptr, HF_ERR_NO_DEVICE, "Error in function detectDevices: device %d of %d", cfg->device_index, ndev);

    hf_ctx* c = new (std::nothrow) hf_ctx();
    if (!c) return fail(nullptr, HF_ERR_OUT_OF_MEMORY, "hf_create: host allocation failed");
    c->cfg = *cfg;
    c->device = cfg->device_index >= 0 ? cfg->device_index : 0;   // (-1: settled by detectDevices below)
    hf::Geom& g = c->g;
    g.hdr = cfg->is_hdr ? 1 : 0;
    g.H = cfg->frame_height;
    g.W = cfg->frame_width;
    g.in_stride = cfg->input_stride > 0 ? cfg->input_stride : g.W;     // opticalFlowCalcSDR.cpp:212
    g.out_stride = cfg->output_stride > 0 ? cfg->output_stride : g.W;  // :213
    g.rs = 0;
    while ((g.H >> g.rs) > cfg->max_calc_res) g.rs++;                  // :217-220
    g.lw = (int)std::ceil((double)g.W / std::pow(2.0, g.rs));          // :221
    g.lh = (int)std::ceil((double)g.H / std::pow(2.0, g.rs));          // :222
    if (g.in_stride < g.W || g.out_stride < g.W) {
        delete c;
        return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_create: strides must be >= frame width");
    }
    if (c->cfg.blur_radius <= 0) c->cfg.blur_radius = 4;               // blurFlowKernelSDR.h:4
    if (c->cfg.blur_radius > 64) { delete c; return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_create: blur_radius must be <= 64"); }
    c->p.delta_scalar = cfg->delta_scalar;
    c->p.neighbor_scalar = cfg->neighbor_scalar;
    c->p.black_level = cfg->black_level;
    c->p.white_level = cfg->white_level;
    c->p.search_radius = cfg->search_radius > 0 ? cfg->search_radius : kMinSearchRadius;  // :216
    c->p.frame_count = 0;
    if (c->p.search_radius < 2 || c->p.search_radius > kMaxSearchRadius) {
        delete c;
        return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_create: search_radius must be in [2, 16]");
    }

    const size_t bpp = g.hdr ? 2 : 1;
    c->in_bytes = bpp * ((size_t)g.H * g.in_stride + (size_t)(g.H / 2) * g.in_stride);     // :20
    c->out_bytes = bpp * ((size_t)g.H * g.out_stride + (size_t)(g.H / 2) * g.out_stride);  // :33
    c->plane_elems = (size_t)g.lw * g.lh;
    const int ws0 = initial_window(g.lw, g.lh);
    const int max_iters = ilog2(ws0);
    c->pl = hf::make_phase_layout(g, max_iters);
    {
        size_t elems = 0;
        for (int k = 0, w = ws0; k < max_iters; k++, w >>= 1) {
            hf::FlowLevel L{};
            L.window = w; L.log2w = ilog2(w);
            L.nwx = (g.lw + w - 1) / w; L.nwy = (g.lh + w - 1) / w;
            c->levels.push_back(L);
            elems += 2 * (size_t)L.nwx * L.nwy;
        }
        c->tables_bytes = (elems + 8) * sizeof(int16_t);
    }
    // window sums: only levels with windows > 32 use them, at most ceil(lw/64)*ceil(lh/64) windows each
    const size_t nwin_max = (size_t)((g.lw + 63) / 64) * ((g.lh + 63) / 64) + 1;
    c->sums_stride = nwin_max * 16;
    c->sums_bytes = (size_t)kMaxSteps * c->sums_stride * sizeof(uint32_t);
    int rc = HF_OK;
    auto bail = [&](int code) { std::string e = c->err; hf_destroy(c); g_create_error = e; return code; };
    {   // detectDevices (opticalFlowCalc.cpp:45-109): the device must offer the memory, LDS and workgroup size the calculator
        // needs.  The reference prices 9 H S_in + 3 H S_out (HDR worst case) + offset / sum arrays against the device's TOTAL
        // memory and takes the FIRST device that qualifies; this build keeps 3 frames + 3 phase planes + 1 output frame + small
        // tables, priced exactly, and additionally requires that much memory to be FREE on the device it settles on.
        // device_index >= 0 pins the ordinal (one process per GPU); -1 scans like the reference.
        const uint64_t required = 3 * c->in_bytes + 3 * c->pl.bytes + c->out_bytes + c->tables_bytes + c->sums_bytes +
                                  2 * c->plane_elems * sizeof(int16_t) * 3 + 2 * c->plane_elems * sizeof(uint32_t);
        std::vector<hf_device_caps> caps((size_t)ndev);
        for (int d = 0; d < ndev; d++) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, d) != hipSuccess) { caps[(size_t)d] = hf_device_caps{}; continue; }
            caps[(size_t)d] = hf_device_caps{(uint64_t)prop.totalGlobalMem, (uint64_t)prop.sharedMemPerBlock, prop.maxThreadsPerBlock, prop.warpSize};
        }
        char why[384] = "";
        const int first = cfg->device_index >= 0 ? cfg->device_index : 0, last = cfg->device_index >= 0 ? cfg->device_index + 1 : ndev;
        int chosen = -1;
        for (int start = first; start < last && chosen < 0;) {
            const int rel = hf_select_device(caps.data() + start, last - start, required, why, sizeof(why));
            if (rel < 0) break;
            const int d = start + rel;
            size_t free_b = 0, total_b = 0;
            if (hipSetDevice(d) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= required) { chosen = d; break; }
            snprintf(why, sizeof(why), "Not enough VRAM available! Required: %llu MB, Available: %llu MB", (unsigned long long)(required / 1024 / 1024),
                     (unsigned long long)(free_b / 1024 / 1024));
            start = d + 1;   // suitable on paper but occupied: move on, as the reference's loop does for an unsuitable device
        }
        if (chosen < 0) {
            fail(c, HF_ERR_NO_DEVICE, "Error in function detectDevices: no suitable HIP GPU found among device%s %d..%d! %s", last - first > 1 ? "s" : "",
                 first, last - 1, why);
            return bail(HF_ERR_NO_DEVICE);
        }
        c->device = chosen;
        c->cfg.device_index = chosen;
    }
    if ((rc = set_device(c))) return bail(rc);
#define HF_TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) { \
        fail(c, _e == hipErrorOutOfMemory ? HF_ERR_OUT_OF_MEMORY : HF_ERR_HIP, "HIP error %d (%s) in %s", (int)_e, hipGetErrorString(_e), #call); \
        return bail(_e == hipErrorOutOfMemory ? HF_ERR_OUT_OF_MEMORY : HF_ERR_HIP); } } while (0)
    HF_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (cfg->flags & HF_FLAG_DUAL_STREAM) {
        HF_TRY(hipStreamCreateWithFlags(&c->own_warp_stream, hipStreamNonBlocking));
        HF_TRY(hipEventCreateWithFlags(&c->ev_flow[0], hipEventDisableTiming));
        HF_TRY(hipEventCreateWithFlags(&c->ev_flow[1], hipEventDisableTiming));
    }
    c->own_stream = c->stream;
    c->warp_stream = c->own_warp_stream ? c->own_warp_stream : c->stream;
    HF_TRY(hipEventCreateWithFlags(&c->ev_chain_done, hipEventDisableTiming));
    HF_TRY(hipEventCreateWithFlags(&c->ev_warps_done, hipEventDisableTiming));
    for (int i = 0; i < 3; i++) {
        HF_TRY(hipMalloc(&c->ring_store[i], c->in_bytes));
        c->ring[i] = c->ring_store[i];
        HF_TRY(hipMalloc((void**)&c->pp[i], c->pl.bytes));
        HF_TRY(hipMemsetAsync(c->ring[i], 0, c->in_bytes, c->stream));
        HF_TRY(hipMemsetAsync(c->pp[i], 0, c->pl.bytes, c->stream));
    }
    HF_TRY(hipMalloc(&c->out_frame, c->out_bytes));
    HF_TRY(hipMemsetAsync(c->out_frame, 0, c->out_bytes, c->stream));
    c->out_target = c->out_frame;
    HF_TRY(hipMalloc((void**)&c->tables, c->tables_bytes));
    HF_TRY(hipMemsetAsync(c->tables, 0, c->tables_bytes, c->stream));
    {
        size_t o = 0;
        for (auto& L : c->levels) {
            L.tx = c->tables + o; o += (size_t)L.nwx * L.nwy;
            L.ty = c->tables + o; o += (size_t)L.nwx * L.nwy;
        }
    }
    HF_TRY(hipMalloc((void**)&c->off_view, 2 * c->plane_elems * sizeof(int16_t)));
    for (int i = 0; i < 2; i++) {
        HF_TRY(hipMalloc((void**)&c->blurred[i], 2 * c->plane_elems * sizeof(int16_t)));
        HF_TRY(hipMemsetAsync(c->blurred[i], 0, 2 * c->plane_elems * sizeof(int16_t), c->stream));
        HF_TRY(hipMalloc((void**)&c->blurred_xy[i], c->plane_elems * sizeof(uint32_t)));
        HF_TRY(hipMemsetAsync(c->blurred_xy[i], 0, c->plane_elems * sizeof(uint32_t), c->stream));
    }
    HF_TRY(hipMalloc((void**)&c->sums, c->sums_bytes));
    HF_TRY(hipMemsetAsync(c->sums, 0, c->sums_bytes, c->stream));
    HF_TRY(hipMalloc((void**)&c->d_probe, 64 * sizeof(float)));
    // m_totalFrameDelta is stored by the chain straight into mapped pinned memory (reference: blocking 4-byte
    // readback in the middle of the chain, opticalFlowCalcSDR.cpp:91-94)
    HF_TRY(hipHostMalloc((void**)&c->h_total_delta, 64, hipHostMallocMapped));
    *c->h_total_delta = 0;
    HF_TRY(hipHostGetDevicePointer((void**)&c->d_total_delta, c->h_total_delta, 0));
    HF_TRY(hipEventCreate(&c->ev_upload));
    HF_TRY(hipEventCreate(&c->ev_flow_end));
    HF_TRY(hipEventCreate(&c->ev_warp_start));
    HF_TRY(hipEventCreate(&c->ev_warp_end));
    HF_TRY(hipEventCreate(&c->ev_user0));
    HF_TRY(hipEventCreate(&c->ev_user1));
    HF_TRY(hipStreamSynchronize(c->stream));
#undef HF_TRY
    *out_ctx = c;
    return HF_OK;
}

void hf_destroy(hf_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) { leave_warp_stream(c); hipStreamSynchronize(c->stream); }  // clFinish (opticalFlowCalcSDR.cpp:186)
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
    for (int i = 0; i < 3; i++) { if (c->ring_store[i]) hipFree(c->ring_store[i]); if (c->pp[i]) hipFree(c->pp[i]); }
    if (c->tables) hipFree(c->tables);
    if (c->off_view) hipFree(c->off_view);
    if (c->out_frame) hipFree(c->out_frame);
    for (int i = 0; i < 2; i++) { if (c->blurred[i]) hipFree(c->blurred[i]); if (c->blurred_xy[i]) hipFree(c->blurred_xy[i]); }
    if (c->sums) hipFree(c->sums);
    if (c->d_probe) hipFree(c->d_probe);
    if (c->h_total_delta) hipHostFree(c->h_total_delta);
    for (auto& sp : c->spans) { hipEventDestroy(sp.b); hipEventDestroy(sp.e); }
    for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
    if (c->io_in) { hipStreamSynchronize(c->io_in); hipStreamSynchronize(c->io_out); hipStreamDestroy(c->io_in); hipStreamDestroy(c->io_out); }
    for (int i = 1; i < hf_ctx::kOutRing; i++) if (c->out_ring[i]) hipFree(c->out_ring[i]);
    for (hipEvent_t e : {c->ev_h2d, c->ev_last_launch, c->ev_out_ready}) if (e) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_slot_prep) if (e) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_d2h) if (e) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_dl) if (e) hipEventDestroy(e);
    if (c->ev_flow_done) hipEventDestroy(c->ev_flow_done);
    for (hipEvent_t e : c->ev_flow) if (e) hipEventDestroy(e);
    if (c->ev_chain_done) hipEventDestroy(c->ev_chain_done);
    if (c->ev_warps_done) hipEventDestroy(c->ev_warps_done);
    hipEvent_t evs[] = {c->ev_upload, c->ev_flow_end, c->ev_warp_start, c->ev_warp_end, c->ev_user0, c->ev_user1};
    for (hipEvent_t e : evs) if (e) hipEventDestroy(e);
    if (c->own_warp_stream) hipStreamDestroy(c->own_warp_stream);
    if (c->own_stream) hipStreamDestroy(c->own_stream);   // c->stream may be a batch's shared stream (not ours)
    delete c;
}

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

int hf_update_frame_async(hf_ctx* c, const void* pinned_host_frame) {
    HF_CHECK_CTX(c);
    if (!pinned_host_frame) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_update_frame_async: null frame");
    if (c->batch) return fail(c, HF_ERR_STATE, "hf_update_frame_async: the context is a member of a batch (asynchronous host I/O uses side streams of its own)");
    if (int rc = set_device(c)) return rc;
    if (int rc = io_init(c)) return rc;
    // the slot about to be overwritten holds the oldest frame: its last readers are the warp/copy launches
    // issued so far and its own (three updates old) phase-plane build
    if (c->have_last_launch) HF_HIP(c, hipStreamWaitEvent(c->io_in, c->ev_last_launch, 0));
    HF_HIP(c, hipStreamWaitEvent(c->io_in, c->ev_slot_prep[0], 0));
    HF_HIP(c, hipMemcpyAsync(c->ring_store[0], pinned_host_frame, c->in_bytes, hipMemcpyHostToDevice, c->io_in));
    HF_HIP(c, hipEventRecord(c->ev_h2d, c->io_in));
    if (int rc = leave_warp_stream(c)) return rc;
    if (c->timing()) {
        HF_HIP(c, hipEventRecord(c->ev_upload, c->stream));
        c->upload_recorded = true;
    }
    HF_HIP(c, hipStreamWaitEvent(c->stream, c->ev_h2d, 0));
    c->ring[0] = c->ring_store[0];
    hf::launch_prep_frame(c->g, c->pl, c->ring[0], c->pp[0], c->stream);
    c->plane_pending[0] = false;
    HF_HIP(c, hipGetLastError());
    HF_HIP(c, hipEventRecord(c->ev_slot_prep[0], c->stream));
    rotate_after_upload(c);
    return HF_OK;
}

int hf_download_frame_async(hf_ctx* c, void* pinned_host_out) {
    HF_CHECK_CTX(c);
    if (!pinned_host_out) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_download_frame_async: null buffer");
    if (c->batch) return fail(c, HF_ERR_STATE, "hf_download_frame_async: the context is a member of a batch (asynchronous host I/O uses side streams of its own)");
    if (int rc = set_device(c)) return rc;
    if (int rc = io_init(c)) return rc;
    hipStream_t last = c->on_warp_stream ? c->warp_stream : c->stream;   // where the frame was just produced
    HF_HIP(c, hipEventRecord(c->ev_out_ready, last));
    HF_HIP(c, hipStreamWaitEvent(c->io_out, c->ev_out_ready, 0));
    HF_HIP(c, hipMemcpyAsync(pinned_host_out, c->out_target, c->out_bytes, hipMemcpyDeviceToHost, c->io_out));
    HF_HIP(c, hipEventRecord(c->ev_dl[c->dl_issued % hf_ctx::kDlRing], c->io_out));
    c->dl_issued++;
    for (int i = 0; i < hf_ctx::kOutRing; i++)
        if (c->out_target == c->out_ring[i]) {   // internal output: the next frame goes to the next ring slot
            HF_HIP(c, hipEventRecord(c->ev_d2h[i], c->io_out));
            c->d2h_pending[i] = true;
            c->out_idx = (i + 1) % hf_ctx::kOutRing;
            c->out_target = c->out_ring[c->out_idx];
            break;
        }
    c->warp_started = false;
    return HF_OK;
}

int hf_update_frame_device_ref(hf_ctx* c, const void* device_frame) {
    HF_CHECK_CTX(c);
    if (!device_frame) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_update_frame_device_ref: null frame");
    return update_common(c, device_frame, hipMemcpyDeviceToDevice, true);
}

static int check_flow_params(hf_ctx* c) {
    const int R = c->p.search_radius;
    if (R < 2 || R > kMaxSearchRadius) return fail(c, HF_ERR_INVALID_ARGUMENT, "calculateOpticalFlow: search radius %d outside [2, 16]", R);
    if (c->p.delta_scalar < 0 || c->p.delta_scalar > 24 || c->p.neighbor_scalar < 0 || c->p.neighbor_scalar > 24)
        return fail(c, HF_ERR_INVALID_ARGUMENT, "calculateOpticalFlow: delta/neighbor scalar outside [0, 24]");
    return HF_OK;
}

// Bookkeeping behind an enqueued chain (eager, graph replay or batch): timing event, flow buffer swap.
static int after_flow_enqueued(hf_ctx* c, hipStream_t s) {
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

int hf_calculate_optical_flow(hf_ctx* c) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    if (int rc = check_flow_params(c)) return rc;
    if (int rc = leave_warp_stream(c)) return rc;
    if (ensure_older_planes(&c, 1, c->stream)) return fail(c, HF_ERR_HIP, "phase-plane launch failed");

    int span = -1;
    if (c->cfg.flags & HF_FLAG_NO_GRAPH) {
        span = span_begin(c, 2);
        if (int rc = enqueue_flow_chain(c)) return rc;
    } else {
        const auto key = std::make_tuple(c->ring_phase, c->blur_phase, c->p.search_radius, c->p.delta_scalar, c->p.neighbor_scalar);
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

// ---- batches (throughput drivers; include/hopperflow.h) ----
struct hf_batch {
    std::vector<hf_ctx*> members;
    std::vector<hipStream_t> own_streams;   // the members' own streams, restored by hf_batch_destroy
    std::vector<hipStream_t> own_warp_streams;
    std::vector<hipStream_t> warp_streams;  // HF_FLAG_DUAL_STREAM members: shared streams their warps are issued on
    hipStream_t stream = nullptr;           // = members[0]'s stream, shared by all members while the batch exists
    std::map<std::vector<int>, hipGraphExec_t> graphs;
    bool defer_planes = false;              // hf_batch_run_period: grid samples at update, full plane of frame N-1 from the warp launch
    std::string err;
};

static thread_local std::string g_batch_error;
static int batch_fail(hf_batch* b, int code, const std::string& msg) {
    (b ? b->err : g_batch_error) = "[HopperRender] " + msg;
    return code;
}

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
                          m->cfg.blur_radius == l->cfg.blur_radius;
        if (!same) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_batch_create: members differ in geometry, device, iterations or blur radius");
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
    delete b;
}

static int batch_update(hf_batch* b, const void* const* device_frames, bool defer);
int hf_batch_update_frames_device_ref(hf_batch* b, const void* const* device_frames) { return batch_update(b, device_frames, false); }

// defer: only the grid samples of the new frames now (what the chain of this period reads of them); their full planes are built by
// the next period's warp launch or, failing that, by ensure_older_planes
static int batch_update(hf_batch* b, const void* const* device_frames, bool defer) {
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
static int batch_check_flow_params(hf_batch* b) {
    hf_ctx* l = b->members[0];
    for (hf_ctx* m : b->members) {
        if (int rc = check_flow_params(m)) return batch_fail(b, rc, m->err);
        if (m->p.search_radius != l->p.search_radius || m->p.delta_scalar != l->p.delta_scalar || m->p.neighbor_scalar != l->p.neighbor_scalar)
            return batch_fail(b, HF_ERR_INVALID_ARGUMENT, "hf_batch_calculate_optical_flow: members differ in search radius / delta / neighbor scalar");
    }
    return HF_OK;
}

int hf_batch_calculate_optical_flow(hf_batch* b) {
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
    hf_ctx* l = b->members[0];
    const int n = (int)b->members.size();
    if (hipSetDevice(l->device) != hipSuccess) return batch_fail(b, HF_ERR_HIP, "hipSetDevice failed");
    if (int rc = batch_check_flow_params(b)) return rc;
    std::vector<int> key = {l->p.search_radius, l->p.delta_scalar, l->p.neighbor_scalar};
    for (hf_ctx* m : b->members) {
        if (int rc = leave_warp_stream(m)) return batch_fail(b, rc, m->err);
        key.push_back(m->ring_phase * 2 + m->blur_phase);
    }
    if (ensure_older_planes(b->members.data(), n, b->stream)) return batch_fail(b, HF_ERR_HIP, "phase-plane launch failed");
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

// Fills the period descriptor of one context: frames N-2 / N-1, the PREVIOUS flow (:154-156), levels, outputs.
// flow_index 1: the period is issued BEFORE the chain of its source period -- the previous flow is still the newest one
static void fill_period(hf_ctx* c, int n, const float* t, void* const* outs, hf::WarpPeriod& p, int flow_index = 0) {
    const float scale = c->g.hdr ? 256.0f : 1.0f;
    p.frame12 = c->ring[0]; p.frame21 = c->ring[1];
    p.flow = c->blurred[flow_index]; p.flow_xy = c->blurred_xy[flow_index];
    p.black = c->p.black_level * scale; p.white = c->p.white_level * scale;
    p.n_out = n;
    for (int i = 0; i < n; i++) { p.ts[i] = t[i]; p.outs[i] = outs[i] ? outs[i] : c->out_frame; }
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

static int batch_interpolate(hf_batch* b, const int* n_out, const float* t, void* const* device_out, int mode, bool before_chain, bool* launched);
int hf_batch_interpolate_period(hf_batch* b, const int* n_out, const float* t, void* const* device_out, int mode) {
    return batch_interpolate(b, n_out, t, device_out, mode, false, nullptr);
}

// before_chain (hf_batch_run_period with deferred phase planes): the period's warps go out AHEAD of the period's chain -- they read
// frames N-2 / N-1 and the previous flow, which the chain does not touch -- and build the full plane of frame N-1 that the chain
// then reads.  Only the one-launch path qualifies; *launched = false means nothing was enqueued and the caller keeps the usual order.
static int batch_interpolate(hf_batch* b, const int* n_out, const float* t, void* const* device_out, int mode, bool before_chain, bool* launched) {
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
        const int span = span_open(l, 0);
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

int hf_batch_run_period(hf_batch* b, const void* const* device_frames, int calculate_flow, const int* n_out, const float* t,
                        void* const* device_out, int mode) {
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
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

int hf_batch_sync(hf_batch* b) {
    if (!b) return batch_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "null batch");
    for (hf_ctx* m : b->members)
        if (int rc = hf_sync(m)) return batch_fail(b, rc, m->err);
    return HF_OK;
}

static int download_common(hf_ctx* c, void* dst, hipMemcpyKind kind) {
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

int hf_sync(hf_ctx* c) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    return sync_ctx(c);
}

int hf_wait_flow(hf_ctx* c) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    if (!c->async()) return HF_OK;                       // blocking contexts have finished every call already
    if (c->batch) return fail(c, HF_ERR_STATE, "hf_wait_flow: the context is a member of a batch (use hf_sync)");
    if (!c->flow_done_recorded) return HF_OK;
    HF_HIP(c, hipEventSynchronize(c->ev_flow_done));
    if (c->delta_pending) { c->total_frame_delta = *c->h_total_delta; c->delta_pending = false; }
    return HF_OK;
}

uint64_t hf_downloads_issued(const hf_ctx* c) { return c ? c->dl_issued : 0; }

int hf_wait_download(hf_ctx* c, uint64_t index) {
    HF_CHECK_CTX(c);
    if (index >= c->dl_issued) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_wait_download: download %llu has not been issued (%llu so far)",
                                           (unsigned long long)index, (unsigned long long)c->dl_issued);
    if (int rc = set_device(c)) return rc;
    // The ring keeps the events of the last kDlRing downloads.  A slot that has been reused belongs to a LATER download of the same
    // in-order stream, so waiting for it (or, for an index older than the whole ring, for the oldest event still kept) implies that
    // download `index` has landed.
    const uint64_t oldest = c->dl_issued > (uint64_t)hf_ctx::kDlRing ? c->dl_issued - (uint64_t)hf_ctx::kDlRing : 0;
    const uint64_t wait_for = index < oldest ? oldest : index;
    HF_HIP(c, hipEventSynchronize(c->ev_dl[wait_for % hf_ctx::kDlRing]));
    return HF_OK;
}

int hf_get_params(const hf_ctx* c, hf_params* out) {
    if (!c || !out) return HF_ERR_INVALID_ARGUMENT;
    *out = c->p;
    return HF_OK;
}

int hf_set_params(hf_ctx* c, const hf_params* in) {
    HF_CHECK_CTX(c);
    if (!in) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_set_params: null");
    c->p = *in;  // validated at use, like the reference (fields are poked directly)
    return HF_OK;
}

int hf_get_stats(hf_ctx* c, hf_stats* out) {
    HF_CHECK_CTX(c);
    if (!out) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_get_stats: null");
    memset(out, 0, sizeof(*out));
    out->total_frame_delta = c->total_frame_delta;
    out->frame_count = c->p.frame_count;
    out->ofc_calc_time = c->ofc_calc_time;
    out->ofc_avg_calc_time = c->ofc_avg;
    out->ofc_peak_calc_time = c->ofc_peak;
    out->warp_calc_time = c->warp_calc_time;
    out->res_scalar = c->g.rs;
    out->low_width = c->g.lw;
    out->low_height = c->g.lh;
    out->frame_width = c->g.W;
    out->frame_height = c->g.H;
    out->input_stride = c->g.in_stride;
    out->output_stride = c->g.out_stride;
    out->iterations = c->last_iterations;
    out->initial_window = c->initial_window;
    out->input_frame_bytes = c->in_bytes;
    out->output_frame_bytes = c->out_bytes;
    out->phase_plane_bytes = c->pl.bytes;
    return HF_OK;
}

int hf_get_profile(hf_ctx* c, hf_profile* out) {
    HF_CHECK_CTX(c);
    if (!out) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_get_profile: null");
    if (int rc = set_device(c)) return rc;
    if (int rc = sync_ctx(c)) return rc;
    *out = c->prof;
    return HF_OK;
}

int hf_set_profile_interval(hf_ctx* c, int warp_every, int flow_every) {
    HF_CHECK_CTX(c);
    if (warp_every < 1 || flow_every < 1) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_set_profile_interval: intervals must be >= 1");
    c->prof_every[0] = c->prof_every[1] = warp_every;
    c->prof_every[2] = flow_every;
    return HF_OK;
}

int hf_reset_profile(hf_ctx* c) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    if (int rc = sync_ctx(c)) return rc;
    c->prof = hf_profile{};
    return HF_OK;
}

int hf_read_offsets(hf_ctx* c, int16_t* host_out) {
    HF_CHECK_CTX(c);
    if (!host_out) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_read_offsets: null");
    if (int rc = set_device(c)) return rc;
    if (int rc = sync_ctx(c)) return rc;
    hf::launch_expand_offsets(c->g, c->last_level, c->off_view, c->stream);
    HF_HIP(c, hipMemcpyAsync(host_out, c->off_view, 2 * c->plane_elems * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    HF_HIP(c, hipStreamSynchronize(c->stream));
    return HF_OK;
}

int hf_read_blurred_flow(hf_ctx* c, int idx, int16_t* host_out) {
    HF_CHECK_CTX(c);
    if (!host_out || idx < 0 || idx > 1) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_read_blurred_flow: bad argument");
    if (int rc = set_device(c)) return rc;
    if (int rc = sync_ctx(c)) return rc;
    HF_HIP(c, hipMemcpyAsync(host_out, c->blurred[idx], 2 * c->plane_elems * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    HF_HIP(c, hipStreamSynchronize(c->stream));
    return HF_OK;
}

int hf_read_phase_plane(hf_ctx* c, int ring_slot, void* host_out, int* complete) {
    HF_CHECK_CTX(c);
    if (!host_out || ring_slot < 0 || ring_slot > 2) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_read_phase_plane: bad argument");
    if (int rc = set_device(c)) return rc;
    if (int rc = sync_ctx(c)) return rc;
    HF_HIP(c, hipMemcpyAsync(host_out, c->pp[ring_slot], c->pl.bytes, hipMemcpyDeviceToHost, c->stream));
    HF_HIP(c, hipStreamSynchronize(c->stream));
    if (complete) *complete = c->plane_pending[ring_slot] ? 0 : 1;
    return HF_OK;
}

int hf_write_blurred_flow(hf_ctx* c, int idx, const int16_t* host_in) {
    HF_CHECK_CTX(c);
    if (!host_in || idx < 0 || idx > 1) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_write_blurred_flow: bad argument");
    if (int rc = set_device(c)) return rc;
    if (int rc = sync_ctx(c)) return rc;
    HF_HIP(c, hipMemcpyAsync(c->blurred[idx], host_in, 2 * c->plane_elems * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
    hf::launch_pack_flow(c->g, c->blurred[idx], c->blurred_xy[idx], c->stream);
    HF_HIP(c, hipStreamSynchronize(c->stream));
    return HF_OK;
}

int hf_timer_begin(hf_ctx* c) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    if (int rc = leave_warp_stream(c)) return rc;
    HF_HIP(c, hipEventRecord(c->ev_user0, c->stream));
    return HF_OK;
}

int hf_timer_end(hf_ctx* c, float* elapsed_ms) {
    HF_CHECK_CTX(c);
    if (!elapsed_ms) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_timer_end: null");
    if (int rc = set_device(c)) return rc;
    if (int rc = leave_warp_stream(c)) return rc;
    HF_HIP(c, hipEventRecord(c->ev_user1, c->stream));
    HF_HIP(c, hipEventSynchronize(c->ev_user1));
    HF_HIP(c, hipEventElapsedTime(elapsed_ms, c->ev_user0, c->ev_user1));
    return HF_OK;
}

int hf_debug_bounds_violations(hf_ctx* c, uint32_t* count, uint32_t first[4], int reset) {
    if (!c || !count) return HF_ERR_INVALID_ARGUMENT;
    if (int rc = set_device(c)) return rc;
    if (hipDeviceSynchronize() != hipSuccess) return fail(c, HF_ERR_HIP, "hf_debug_bounds_violations: hipDeviceSynchronize failed");
    unsigned rec[5] = {0, 0, 0, 0, 0};
    const bool a = hf::dbg_bounds_read_kernels(rec, reset != 0), b = hf::dbg_bounds_read_flow(rec, reset != 0);
    if (!a || !b) return fail(c, HF_ERR_STATE, "hf_debug_bounds_violations: this library was built without -DHF_DEBUG_BOUNDS (python -m hopperrender_amd.build --debug-bounds)");
    *count = rec[0];
    if (first) for (int i = 0; i < 4; i++) first[i] = rec[1 + i];
    return HF_OK;
}

int hf_debug_bounds_selftest(hf_ctx* c) {
    if (!c) return HF_ERR_INVALID_ARGUMENT;
    uint32_t before = 0, after = 0, first[4] = {0, 0, 0, 0};
    if (int rc = hf_debug_bounds_violations(c, &before, nullptr, 0)) return rc;
    int* scratch = nullptr;
    if (hipMalloc(&scratch, 80 * sizeof(int)) != hipSuccess) return fail(c, HF_ERR_OUT_OF_MEMORY, "hf_debug_bounds_selftest: hipMalloc failed");
    hf::launch_bounds_selftest(scratch, c->stream);
    const hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(scratch);
    if (e != hipSuccess) return fail(c, HF_ERR_HIP, "hf_debug_bounds_selftest: launch failed");
    if (int rc = hf_debug_bounds_violations(c, &after, first, 0)) return rc;
    if (after - before != 64u || (before == 0 && first[0] != 999u))
        return fail(c, HF_ERR_STATE, "hf_debug_bounds_selftest: 64 out-of-range indices were issued, %u recorded (first site %u)", after - before, first[0]);
    return HF_OK;
}

int hf_device_rcp(hf_ctx* c, const float* host_in, float* host_out, int n) {
    HF_CHECK_CTX(c);
    if (!host_in || !host_out || n < 1 || n > 32) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_device_rcp: bad argument");
    if (int rc = set_device(c)) return rc;
    if (int rc = sync_ctx(c)) return rc;
    HF_HIP(c, hipMemcpyAsync(c->d_probe, host_in, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hf::launch_rcp_probe(c->d_probe, c->d_probe + 32, n, c->stream);
    HF_HIP(c, hipMemcpyAsync(host_out, c->d_probe + 32, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HF_HIP(c, hipStreamSynchronize(c->stream));
    return HF_OK;
}

int hf_device_malloc(int device_index, size_t bytes, void** out) {
    if (!out) return HF_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device_index) != hipSuccess) return fail(nullptr, HF_ERR_NO_DEVICE, "hf_device_malloc: bad device %d", device_index);
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorOutOfMemory ? HF_ERR_OUT_OF_MEMORY : HF_ERR_HIP, "hf_device_malloc: %s", hipGetErrorString(e));
    return HF_OK;
}

int hf_device_free(int device_index, void* p) {
    if (hipSetDevice(device_index) != hipSuccess) return HF_ERR_NO_DEVICE;
    return hipFree(p) == hipSuccess ? HF_OK : HF_ERR_HIP;
}

int hf_host_malloc_pinned(size_t bytes, void** out) {
    if (!out) return HF_ERR_INVALID_ARGUMENT;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorOutOfMemory ? HF_ERR_OUT_OF_MEMORY : HF_ERR_HIP, "hf_host_malloc_pinned: %s", hipGetErrorString(e));
    return HF_OK;
}

int hf_host_free_pinned(void* p) { return hipHostFree(p) == hipSuccess ? HF_OK : HF_ERR_HIP; }

// hf_memcpy_* have no context, hence no stream of their own, and an extra stream would occupy one of the few hardware
// queues the pair streams need (DESIGN.md "Hardware queues": 50 k -> 42 k frames/s with one more stream alive, 35 k
// with short-lived ones).  They use the legacy stream -- but never while a thread of this process captures a graph: a
// synchronous hipMemcpy there invalidates the capture (HIP error 906, a rare failure of the threads test; 130 errors
// under tools/stress_threads.py).  Captures hold g_capture_mutex shared, these copies exclusively.
static int util_copy(int device_index, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (hipSetDevice(device_index) != hipSuccess) return HF_ERR_NO_DEVICE;
    std::unique_lock<std::shared_mutex> lock(g_capture_mutex);
    return hipMemcpy(dst, src, bytes, kind) == hipSuccess ? HF_OK : HF_ERR_HIP;
}

int hf_memcpy_h2d(int device_index, void* d, const void* h, size_t bytes) { return util_copy(device_index, d, h, bytes, hipMemcpyHostToDevice); }

int hf_memcpy_d2h(int device_index, void* h, const void* d, size_t bytes) { return util_copy(device_index, h, d, bytes, hipMemcpyDeviceToHost); }

}  // extern "C"
