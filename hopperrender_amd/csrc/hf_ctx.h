// hopperrender_amd/csrc/hf_ctx.h -- internal: the context and batch objects behind the C ABI of include/hopperflow.h and the helpers the
// translation units of that ABI share.  The ABI is split by concern:
//   hf_context.hip   context lifetime and state: hf_create / hf_destroy (detectDevices, buffers: opticalFlowCalcSDR.cpp:206-325), parameters,
//                    statistics, profiling spans, parity taps, device-memory helpers
//   hf_calc.hip      the five virtuals of one context: updateFrame, calculateOpticalFlow (the refinement chain as a cached hipGraph), warpFrames,
//                    copyFrame, downloadFrame, and the fused period calls
//   hf_batch.hip     hf_batch: the same calls for up to 32 contexts of one geometry as one set of launches (throughput drivers)
//   hf_async_io.hip  pinned asynchronous H2D / D2H on side streams (hf_update_frame_async / hf_download_frame_async, hf_wait_*)
//
// Host orchestration restated from the reference's opticalFlowCalcSDR.cpp / opticalFlowCalcHDR.cpp (cited per function); the
// architecture differs on purpose:
//   * every uploaded frame is re-laid out once as mirror-padded phase planes, so the candidates of a run
//     of grid pixels are consecutive bytes (hf_flow.hip);
//   * offsets are kept per window in one small table per level, not per pixel;
//   * ONE launch per level (X and Y step fused) for windows <= 32, two per axis for larger windows
//     (reference: fill + calcDeltaSums + determineLowestLayer + adjustOffsetArray = 4 enqueues per step,
//     2.6-8.3 MB of fills);
//   * m_totalFrameDelta is produced on the device and copied to pinned memory inside the graph
//     (reference: blocking 4-byte readback in the middle of the chain, opticalFlowCalcSDR.cpp:91-94);
//   * the whole chain replays as one hipGraph keyed by (ring phase, search radius, scalars).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/config.h"
#include "../../include/hopperflow.h"
#include "../../include/hopperflow_diag.h"
#include "hf_kernels.h"

namespace hfi {

constexpr int kMinSearchRadius = MIN_SEARCH_RADIUS;    // include/config.h (reference config.h:8)
constexpr int kMaxSearchRadius = MAX_SEARCH_RADIUS;    // config.h:9
constexpr int kCalcTimeInterval = CALC_TIME_INTERVAL;  // config.h:17
static_assert(kMaxSearchRadius <= 16, "the chain kernels keep 16 candidates per step in registers");
constexpr int kMaxSteps = 32;          // 2 * log2(max window)

extern std::shared_mutex g_capture_mutex;   // shared: a stream capture is in progress; exclusive: a legacy-stream copy (util_copy)
}  // namespace hfi

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
    uint32_t* sadtab = nullptr;                        // SAD tables (hf_flow.hip): [2][sad_nby][sad_nbx][8] u16 pairs; nullptr: HF_FLAG_NO_SAD_REUSE
    size_t sadtab_bytes = 0;
    int sad_nbx = 0, sad_nby = 0;
    // Content hint of the SAD tables: how many windows of the 32-level chose d = 0 on both axes in the last chains (device counter, published
    // by the chain's last kernel into h_total_delta[1]); below ~30 % of them the tables cost more than they return and the chain runs without
    bool tab_mode = true;                              // the next chain keeps SAD tables
    float still_share = -1.f;                          // smoothed share of such windows (< 0: no chain has reported yet)
    uint32_t* still_count = nullptr;
    uint32_t* counters = nullptr;                      // diagnostic counters (hf_debug_counters_enable), device, hf::kCounterWords u32
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

    std::map<std::tuple<int, int, int, int, int, int>, hipGraphExec_t> graphs;

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

// Per-launch device timestamps of a batch (hf_batch_timeline_*): the start / stop events of every dispatch the batch issues while it is on.
struct hf_timeline final : hf::LaunchObserver {
    struct Rec { const char* name; hipEvent_t b, e; int period; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> events;      // 2 per record, created when the timeline is switched on
    size_t capacity = 0;
    int skip = 0;                        // hf_batch_run_period calls still to pass unobserved before the recording starts
    bool active = false;                 // records left: hf_batch_run_period observes its launches and issues the chain eagerly (no graph replay)
    int period = 0;                      // hf_batch_run_period calls since it was switched on
    uint64_t dropped = 0;                // launches issued while on but out of records (cannot happen: it switches itself off when full)
    bool next(const char* name, hipEvent_t* s, hipEvent_t* e) override {
        if (recs.size() >= capacity) { dropped++; return false; }
        *s = events[2 * recs.size()]; *e = events[2 * recs.size() + 1];
        recs.push_back(Rec{name, *s, *e, period});
        return true;
    }
};

// ---- batches (throughput drivers; include/hopperflow.h) ----
struct hf_batch {
    hf_timeline tl;
    std::vector<hf_ctx*> members;
    std::vector<hipStream_t> own_streams;   // the members' own streams, restored by hf_batch_destroy
    std::vector<hipStream_t> own_warp_streams;
    std::vector<hipStream_t> warp_streams;  // HF_FLAG_DUAL_STREAM members: shared streams their warps are issued on
    hipStream_t stream = nullptr;           // = members[0]'s stream, shared by all members while the batch exists
    std::map<std::vector<int>, hipGraphExec_t> graphs;
    bool defer_planes = false;              // hf_batch_run_period: grid samples at update, full plane of frame N-1 from the warp launch
    std::string err;
};

namespace hfi {

// hf_context.hip
int fail(hf_ctx* c, int code, const char* fmt, ...);
int set_device(hf_ctx* c);
int initial_window(int lw, int lh);
int ilog2(int v);
int effective_iterations(const hf_ctx* c);
hipEvent_t pool_event(hf_ctx* c);
int span_begin(hf_ctx* c, int kind, hipStream_t stream = nullptr);
int span_open(hf_ctx* c, int kind);
void span_end(hf_ctx* c, int idx);
void collect_spans(hf_ctx* c);
int sync_ctx(hf_ctx* c);
int util_copy(int device_index, void* dst, const void* src, size_t bytes, hipMemcpyKind kind);

// hf_calc.hip
int enqueue_flow_chain(hf_ctx* const* cs, int n, hipStream_t s);
int enqueue_flow_chain(hf_ctx* c);
bool choose_tab_mode(hf_ctx* const* cs, int n);
int ensure_older_planes(hf_ctx* const* cs, int n, hipStream_t s);
void finish_flow_timing(hf_ctx* c);
int enter_warp_stream(hf_ctx* c);
int leave_warp_stream(hf_ctx* c);
int rotate_after_upload(hf_ctx* c);
int update_common(hf_ctx* c, const void* src, hipMemcpyKind kind, bool by_reference = false);
int check_flow_params(hf_ctx* c);
int after_flow_enqueued(hf_ctx* c, hipStream_t s);
void fill_period(hf_ctx* c, int n, const float* t, void* const* outs, hf::WarpPeriod& p, int flow_index = 0);
int download_common(hf_ctx* c, void* dst, hipMemcpyKind kind);

// hf_batch.hip
int batch_fail(hf_batch* b, int code, const std::string& msg);
int batch_update(hf_batch* b, const void* const* device_frames, bool defer);
int batch_check_flow_params(hf_batch* b);
int batch_interpolate(hf_batch* b, const int* n_out, const float* t, void* const* device_out, int mode, bool before_chain, bool* launched);

// hf_async_io.hip
int io_init(hf_ctx* c);
int guard_output_slot(hf_ctx* c, const void* target, hipStream_t launch_stream);
int note_launch(hf_ctx* c, hipStream_t launch_stream);

}  // namespace hfi

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
