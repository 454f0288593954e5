// hopperrender_amd/csrc/hf_hostio.cpp -- the host-I/O driver of one rank and the timeline planner, plain host C++ on top of
// the C ABI (include/hopperflow.h "Streaming host-I/O driver"); no HIP here.
//
// The reference's filter moves every frame through host memory with blocking transfers on its streaming thread (updateFrame,
// opticalFlowCalcSDR.cpp:19-29; downloadFrame, :31-42) and decides per output frame between warpFrames and copyFrame from the
// m_totalFrameDelta history (HopperRender.cpp:938-1197).  A throughput host that converts a clip on several GPUs gives every
// rank (= process = GPU) a contiguous chunk of the source timeline and runs, per rank, this loop:
//
//     pinned input ring  --hf_update_frame_async (H2D side stream)-->  3-frame ring + phase plane
//     chain on the context's stream, warps on its second stream (HF_FLAG_DUAL_STREAM)
//     output frames      --hf_download_frame_async (D2H side stream)--> pinned output ring --> sink(index, frame), in index order
//
// with the filter's protocol state kept by hf_filter.  The only host wait inside a period is hf_wait_flow: the decision warp vs
// copy needs THIS period's frame delta.  An output slot is drained (hf_wait_download) right before it is reused.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/config.h"
#include "../../include/hopperflow.h"

struct hf_hostio {
    hf_ctx* ctx = nullptr;
    hf_hostio_config cfg{};
    hf_stats st{};
    std::vector<void*> ins, outs;
    uint64_t bytes_in = 0, bytes_out = 0;
    std::string err;
};

namespace {

thread_local std::string g_hostio_error;

int hio_fail(hf_hostio* h, int code, const std::string& msg) {
    (h ? h->err : g_hostio_error) = "[HopperRender] " + msg;
    return code;
}

}  // namespace

extern "C" {

const char* hf_hostio_last_error(const hf_hostio* h) { return h ? h->err.c_str() : g_hostio_error.c_str(); }

// The filter's schedule for a whole clip, cut into `world` contiguous chunks (HopperRender.cpp:944-948,1192-1197 through
// hf_filter): chunk `rank` owns n_periods source periods from first_period on and needs the frames from first_frame on --
// `overlap` frames so that the ring holds N-2, N-1, N and the previous flow exists, plus `delta_history` periods whose flow is
// replayed so that the scene-change decisions equal the sequential run's (HopperRender.cpp:1131-1144: the average of up to 10
// earlier periods + the current + the next one -> 12).  Periods before the third frame of the clip only copy frames (:955,1179).
int hf_shard_timeline(int64_t n_source_frames, int world, int rank, int64_t source_frame_time, int64_t target_frame_time, int overlap,
                      int delta_history, hf_timeline_chunk* out, int32_t* n_out, float* t, int64_t t_capacity) {
    if (!out || n_source_frames < 0 || world < 1 || rank < 0 || rank >= world || overlap < 0 || delta_history < 0)
        return hio_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_shard_timeline: bad argument");
    hf_filter_config fc{};
    fc.struct_size = sizeof(fc);
    fc.scene_change_threshold = -1;
    fc.source_frame_time = source_frame_time;
    fc.target_frame_time = target_frame_time;
    fc.frame_output_mode = HF_MODE_BLENDED_FRAME;
    fc.active = 1;
    hf_filter* f = nullptr;
    if (int rc = hf_filter_create(&fc, &f)) return hio_fail(nullptr, rc, "hf_shard_timeline: hf_filter_create failed");
    const int64_t base = n_source_frames / world, rem = n_source_frames % world;
    const int64_t start = rank * base + (rank < rem ? rank : rem), count = base + (rank < rem ? 1 : 0);
    hf_timeline_chunk ch{};                                  // (*out is only written on success)
    ch.first_period = start;
    ch.n_periods = count;
    ch.first_frame = start - overlap - delta_history > 0 ? start - overlap - delta_history : 0;
    ch.n_frames = start + count - ch.first_frame;
    int64_t outputs_before = 0, mine = 0;
    int rc = HF_OK;
    for (int64_t k = 0; k < start + count; k++) {
        if (k == start) ch.blend_at_start = hf_filter_blending_scalar(f);
        const int n = hf_filter_begin_source_frame(f);
        if (k >= start && n_out) n_out[k - start] = n;
        for (int i = 0; i < n; i++) {
            if (k >= start) {
                if (t) {
                    if (mine >= t_capacity) { rc = HF_ERR_INVALID_ARGUMENT; break; }
                    t[mine] = (float)hf_filter_blending_scalar(f);
                }
                mine++;
            } else {
                outputs_before++;
            }
            hf_filter_advance_blending_scalar(f);
        }
        if (rc) break;
    }
    hf_filter_destroy(f);
    if (rc) return hio_fail(nullptr, rc, "hf_shard_timeline: t_capacity too small");
    ch.first_output = outputs_before;
    ch.n_outputs = mine;
    *out = ch;
    return HF_OK;
}

int hf_hostio_create(hf_ctx* ctx, const hf_hostio_config* cfg, hf_hostio** out) {
    if (!ctx || !out) return hio_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_hostio_create: null argument");
    *out = nullptr;
    if (cfg && cfg->struct_size != sizeof(hf_hostio_config)) return hio_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_hostio_create: hf_hostio_config.struct_size");
    hf_hostio* h = new (std::nothrow) hf_hostio();
    if (!h) return hio_fail(nullptr, HF_ERR_OUT_OF_MEMORY, "hf_hostio_create: host allocation failed");
    h->ctx = ctx;
    if (cfg) {
        h->cfg = *cfg;
    } else {   // no configuration = the filter's defaults: blended output (m_iFrameOutput, HopperRender.cpp:129), default scene-change threshold
        h->cfg.frame_output_mode = HF_MODE_BLENDED_FRAME;
        h->cfg.scene_change_threshold = -1;
    }
    h->cfg.struct_size = sizeof(hf_hostio_config);
    if (h->cfg.in_ring <= 0) h->cfg.in_ring = 3;
    if (h->cfg.out_ring <= 0) h->cfg.out_ring = 12;
    if (h->cfg.source_frame_time <= 0) h->cfg.source_frame_time = 417083;   // 23.976 fps (HopperRender.cpp:162)
    if (h->cfg.target_frame_time <= 0) h->cfg.target_frame_time = 166667;   // 60 fps (:163)
    if (h->cfg.in_ring < 3 || h->cfg.out_ring < 2) { delete h; return hio_fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_hostio_create: in_ring >= 3 and out_ring >= 2"); }
    if (int rc = hf_get_stats(ctx, &h->st)) { delete h; return hio_fail(nullptr, rc, "hf_hostio_create: hf_get_stats failed"); }
    for (int i = 0; i < h->cfg.in_ring + h->cfg.out_ring; i++) {
        void* p = nullptr;
        const size_t bytes = i < h->cfg.in_ring ? (size_t)h->st.input_frame_bytes : (size_t)h->st.output_frame_bytes;
        if (int rc = hf_host_malloc_pinned(bytes, &p)) { hf_hostio_destroy(h); return hio_fail(nullptr, rc, "hf_hostio_create: pinned allocation failed"); }
        (i < h->cfg.in_ring ? h->ins : h->outs).push_back(p);
    }
    *out = h;
    return HF_OK;
}

void hf_hostio_destroy(hf_hostio* h) {
    if (!h) return;
    if (h->ctx) hf_sync(h->ctx);
    for (void* p : h->ins) hf_host_free_pinned(p);
    for (void* p : h->outs) hf_host_free_pinned(p);
    delete h;
}

int hf_hostio_get_traffic(const hf_hostio* h, uint64_t* bytes_in, uint64_t* bytes_out) {
    if (!h) return HF_ERR_INVALID_ARGUMENT;
    if (bytes_in) *bytes_in = h->bytes_in;
    if (bytes_out) *bytes_out = h->bytes_out;
    return HF_OK;
}

int hf_hostio_run(hf_hostio* h, const hf_timeline_chunk* chunk, const int32_t* n_out, const float* t, hf_hostio_fill_fn fill,
                  hf_hostio_sink_fn sink, void* user, int32_t* kinds) {
    if (!h || !chunk || !fill || !sink || (chunk->n_periods > 0 && (!n_out || !t)))
        return hio_fail(h, HF_ERR_INVALID_ARGUMENT, "hf_hostio_run: null argument");
    hf_ctx* c = h->ctx;
    hf_filter_config fc{};
    fc.struct_size = sizeof(fc);
    fc.scene_change_threshold = h->cfg.scene_change_threshold;
    fc.source_frame_time = h->cfg.source_frame_time;
    fc.target_frame_time = h->cfg.target_frame_time;
    fc.frame_output_mode = h->cfg.frame_output_mode;
    fc.active = 1;
    hf_filter* f = nullptr;
    if (int rc = hf_filter_create(&fc, &f)) return hio_fail(h, rc, "hf_hostio_run: hf_filter_create failed");
    struct Guard { hf_filter* f; ~Guard() { hf_filter_destroy(f); } } guard{f};
#define HIO(call)                                                                                  \
    do {                                                                                           \
        if (int rc_ = (call)) return hio_fail(h, rc_, std::string(#call " failed: ") + hf_last_error(c)); \
    } while (0)
    hf_params p{};
    HIO(hf_get_params(c, &p));
    p.frame_count = 0;                                   // a chunk starts like a new segment (HopperRender.cpp:840)
    HIO(hf_set_params(c, &p));
    const int R = (int)h->outs.size();
    const uint64_t base = hf_downloads_issued(c);
    std::vector<int32_t> kind_of;                        // 1 warp / 0 copy per output of this chunk
    int64_t issued = 0, drained = 0, t_at = 0;
    auto drain = [&](int64_t upto) -> int {
        while (drained < upto) {
            if (int rc = hf_wait_download(c, base + (uint64_t)drained)) return hio_fail(h, rc, std::string("hf_wait_download failed: ") + hf_last_error(c));
            if (int rc = sink(user, drained, h->outs[(size_t)(drained % R)], kind_of[(size_t)drained])) return hio_fail(h, HF_ERR_STATE, "hf_hostio_run: sink returned " + std::to_string(rc));
            drained++;
        }
        return HF_OK;
    };
    for (int64_t k = chunk->first_frame; k < chunk->first_frame + chunk->n_frames; k++) {
        void* slot = h->ins[(size_t)((k - chunk->first_frame) % (int64_t)h->ins.size())];
        // (the upload that last used this slot finished before an earlier hf_wait_flow / hf_sync)
        if (int rc = fill(user, k, slot)) return hio_fail(h, HF_ERR_STATE, "hf_hostio_run: fill returned " + std::to_string(rc));
        HIO(hf_update_frame_async(c, slot));
        h->bytes_in += h->st.input_frame_bytes;
        const uint32_t count = (uint32_t)(k + 1);        // the sequential run's m_frameCount at this frame
        HIO(hf_get_params(c, &p));
        if (p.frame_count >= 3) {                        // HopperRender.cpp:955
            HIO(hf_calculate_optical_flow(c));
            HIO(hf_wait_flow(c));                        // m_totalFrameDelta of this period; the side streams keep running
            hf_stats st{};
            HIO(hf_get_stats(c, &st));
            hf_filter_push_frame_delta(f, count, st.total_frame_delta);   // :959-972
        } else {
            HIO(hf_sync(c));                             // the first two frames of a segment: no chain to wait for
        }
        if (k < chunk->first_period) continue;           // warm-up: ring, previous flow, delta history -- no output
        const int n = n_out[k - chunk->first_period];
        for (int i = 0; i < n; i++, t_at++) {
            const int cut = hf_filter_detect_scene_change(f, count);      // :1126-1176
            if (count >= 3 && !cut) {                    // :1179-1183
                HIO(hf_warp_frames(c, t[t_at], h->cfg.frame_output_mode));
                kind_of.push_back(1);
            } else {
                HIO(hf_copy_frame(c));
                kind_of.push_back(0);
            }
            if (int rc = drain(issued - R + 1)) return rc;   // the slot about to be overwritten must have gone to the sink
            HIO(hf_download_frame_async(c, h->outs[(size_t)(issued % R)]));
            issued++;
            h->bytes_out += h->st.output_frame_bytes;
        }
    }
    if (int rc = drain(issued)) return rc;
    HIO(hf_sync(c));
#undef HIO
    if (kinds && !kind_of.empty()) std::memcpy(kinds, kind_of.data(), kind_of.size() * sizeof(int32_t));
    return HF_OK;
}

}  // extern "C"
