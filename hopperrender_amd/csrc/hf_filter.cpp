// hopperrender_amd/csrc/hf_filter.cpp -- the host-side protocol AROUND the calculator, natively and behind the C ABI
// (include/hopperflow.h, "caller protocol"): which calls the reference's filter makes for a source frame and with which
// arguments.  Restated from reference HopperRender/HopperRender.cpp:
//   :944-948    number of output frames of a source period          -> hf_filter_begin_source_frame
//   :1192-1197  blending-scalar schedule                            -> hf_filter_blending_scalar / _advance_
//   :959-972    3-second history of m_totalFrameDelta               -> hf_filter_push_frame_delta
//   :1126-1176  scene-change decision (warp vs copy), 1-second peaks -> hf_filter_detect_scene_change
//   :1438-1463  search-radius governor (autoAdjustSettings)         -> hf_filter_auto_adjust
//   :820-842    NewSegment / UpdateInterpolationStatus              -> hf_filter_new_segment
// and hf_filter_deliver() = one whole DeliverToRenderer (:938-1197) against an hf_ctx, for hosts that are not C++.
// No HIP here: plain host logic (it decides warp vs copy per output frame, i.e. it is part of results parity for clips).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <new>

#include "../../include/config.h"
#include "../../include/hopperflow.h"

struct hf_filter {
    hf_filter_config cfg{};
    int64_t playback_frame_time = 0;      // m_rtCurrPlaybackFrameTime (NewSegment: source frame time / rate)
    int active = 1;                       // m_iIntActiveState == Active
    int num_int_frames = 1;               // m_iNumIntFrames
    double blending_scalar = 0.0;         // m_dBlendingScalar
    double total_warp_duration = 0.0;     // m_dTotalWarpDuration
    struct Delta { uint32_t frame, total; };
    struct Change { uint32_t frame, d1, d2; };
    std::deque<Delta> deltas;             // m_frameDeltaHistory
    std::deque<Change> changes;           // m_sceneChangeDeltaHistory
    uint32_t peak1 = 0, peak2 = 0;        // m_iPeakSceneChangeDelta, m_iPeakSceneChangeDelta2
    int last_average = 0, last_d1 = 0, last_d2 = 0;
};

namespace {

// HopperRender.cpp:820-831 (UpdateInterpolationStatus): interpolation only when the target rate is higher than the
// playback rate; the histories start over
void update_interpolation_status(hf_filter* f) {
    if (f->cfg.active && f->playback_frame_time > f->cfg.target_frame_time) f->active = 1;
    else f->active = 0;
    f->peak1 = f->peak2 = 0;
    f->deltas.clear();
    f->changes.clear();
}

}  // namespace

extern "C" {

int hf_filter_create(const hf_filter_config* cfg, hf_filter** out) {
    if (!cfg || !out || cfg->struct_size != sizeof(hf_filter_config)) return HF_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    hf_filter* f = new (std::nothrow) hf_filter();
    if (!f) return HF_ERR_OUT_OF_MEMORY;
    f->cfg = *cfg;
    if (f->cfg.source_frame_time <= 0) f->cfg.source_frame_time = 417083;            // HopperRender.cpp:162
    if (f->cfg.target_frame_time <= 0) f->cfg.target_frame_time = 166667;            // :163
    if (f->cfg.scene_change_threshold < 0) f->cfg.scene_change_threshold = DEFAULT_SCENE_CHANGE_THRESHOLD;
    if (f->cfg.frame_output_mode < 0 || f->cfg.frame_output_mode > 6) { delete f; return HF_ERR_INVALID_ARGUMENT; }
    f->playback_frame_time = f->cfg.source_frame_time;
    update_interpolation_status(f);
    *out = f;
    return HF_OK;
}

void hf_filter_destroy(hf_filter* f) { delete f; }

// NewSegment (HopperRender.cpp:834-845): playback frame time from the rate, activation status, histories cleared.
// The caller zeroes the calculator's m_frameCount (:840), or passes the context to hf_filter_deliver's wrapper below.
int hf_filter_new_segment(hf_filter* f, double rate) {
    if (!f || !(rate > 0.0)) return HF_ERR_INVALID_ARGUMENT;
    f->playback_frame_time = (int64_t)((double)f->cfg.source_frame_time * (1.0 / rate));
    update_interpolation_status(f);
    return HF_OK;
}

int hf_filter_set_playback_frame_time(hf_filter* f, int64_t playback_frame_time) {
    if (!f || playback_frame_time <= 0) return HF_ERR_INVALID_ARGUMENT;
    f->playback_frame_time = playback_frame_time;
    return HF_OK;
}

int hf_filter_is_active(const hf_filter* f) { return f ? f->active : 0; }

// HopperRender.cpp:944-948
int hf_filter_begin_source_frame(hf_filter* f) {
    if (!f) return HF_ERR_INVALID_ARGUMENT;
    if (f->active) {
        const double per_output = (double)f->cfg.target_frame_time / (double)f->playback_frame_time;
        f->num_int_frames = (int)std::fmax(std::ceil((1.0 - f->blending_scalar) / per_output), 1.0);
    } else {
        f->num_int_frames = 1;
    }
    return f->num_int_frames;
}

double hf_filter_blending_scalar(const hf_filter* f) { return f ? f->blending_scalar : 0.0; }

// HopperRender.cpp:1192-1197
void hf_filter_advance_blending_scalar(hf_filter* f) {
    if (!f || !f->active) return;
    f->blending_scalar += (double)f->cfg.target_frame_time / (double)f->playback_frame_time;
    if (f->blending_scalar >= 1.0) f->blending_scalar -= 1.0;
}

// HopperRender.cpp:1189
void hf_filter_add_warp_duration(hf_filter* f, double warp_calc_time) {
    if (f) f->total_warp_duration += warp_calc_time;
}

// autoAdjustSettings (HopperRender.cpp:1438-1463).  ofc_calc_time = the calculator's m_ofcCalcTime (seconds);
// *search_radius = its m_opticalFlowSearchRadius, adjusted in place.  Returns -1 / 0 / +1 = what was done.
int hf_filter_auto_adjust(hf_filter* f, double ofc_calc_time, int32_t* search_radius) {
    if (!f || !search_radius) return HF_ERR_INVALID_ARGUMENT;
    const double source_frame_time_s = (double)f->playback_frame_time / 10000000.0;
    const double duration = ofc_calc_time + f->total_warp_duration;
    int step = 0;
    if (duration * UPPER_PERF_BUFFER > source_frame_time_s) {          // too slow: fewer candidates
        if (*search_radius > MIN_SEARCH_RADIUS) { (*search_radius)--; step = -1; }
    } else if (duration * LOWER_PERF_BUFFER < source_frame_time_s) {   // capacity left: more candidates
        if (*search_radius < MAX_SEARCH_RADIUS) { (*search_radius)++; step = 1; }
    }
    f->total_warp_duration = 0.0;
    return step;
}

// HopperRender.cpp:959-972: the newest m_totalFrameDelta joins a 3-second window (unsigned frame arithmetic as there)
int hf_filter_push_frame_delta(hf_filter* f, uint32_t frame_count, uint32_t total_frame_delta) {
    if (!f) return HF_ERR_INVALID_ARGUMENT;
    const int frames_in_3s = (int)(3.0 * 10000000.0 / (double)f->cfg.source_frame_time);
    f->deltas.push_back({frame_count, total_frame_delta});
    while (!f->deltas.empty() && (frame_count - f->deltas.front().frame) > (uint32_t)frames_in_3s) f->deltas.pop_front();
    return HF_OK;
}

// HopperRender.cpp:1126-1176: is the period being output a scene change?  (Evaluated once per OUTPUT frame there,
// so the 1-second peak history receives one entry per output frame; kept.)  Returns 1 = copy instead of warp.
int hf_filter_detect_scene_change(hf_filter* f, uint32_t frame_count) {
    if (!f) return HF_ERR_INVALID_ARGUMENT;
    const size_t n = f->deltas.size();
    if (n < 3) return 0;
    const size_t count = n - 2 < 10 ? n - 2 : 10;
    unsigned long long sum = 0;
    for (size_t i = 0; i < count; i++) sum += f->deltas[n - 2 - i].total;
    const int average = (int)(sum / count);
    const int next = (int)f->deltas[n - 1].total, current = (int)f->deltas[n - 2].total;
    const int d1 = current - average, d2 = current - next;
    f->last_average = average; f->last_d1 = d1; f->last_d2 = d2;
    if (d1 > 0) {
        const int frames_in_1s = (int)(1.0 * 10000000.0 / (double)f->cfg.source_frame_time);
        f->changes.push_back({frame_count, (uint32_t)d1, d2 > 0 ? (uint32_t)d2 : 0u});
        while (!f->changes.empty() && (frame_count - f->changes.front().frame) > (uint32_t)frames_in_1s) f->changes.pop_front();
        f->peak1 = f->peak2 = 0;
        for (const auto& c : f->changes)
            if (c.d1 > f->peak1) { f->peak1 = c.d1; f->peak2 = c.d2; }
    }
    const uint32_t thr = (uint32_t)f->cfg.scene_change_threshold;
    return ((uint32_t)d1 >= thr && d1 > 0 && (uint32_t)d2 >= thr && d2 > 0) ? 1 : 0;
}

int hf_filter_get_state(const hf_filter* f, hf_filter_state* out) {
    if (!f || !out) return HF_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    out->num_int_frames = f->num_int_frames;
    out->active = f->active;
    out->blending_scalar = f->blending_scalar;
    out->total_warp_duration = f->total_warp_duration;
    out->playback_frame_time = f->playback_frame_time;
    out->peak_scene_change_delta = f->peak1;
    out->peak_scene_change_delta2 = f->peak2;
    out->frame_delta_history = (uint32_t)f->deltas.size();
    out->scene_change_history = (uint32_t)f->changes.size();
    out->average_frame_delta = f->last_average;
    out->scene_change_delta1 = f->last_d1;
    out->scene_change_delta2 = f->last_d2;
    return HF_OK;
}

// One DeliverToRenderer (HopperRender.cpp:938-1197) against a calculator context with blocking semantics: the number
// of outputs, the governor, updateFrame, calculateOpticalFlow + delta history, then per output frame the scene-change
// decision, warpFrames or copyFrame, downloadFrame, warp-time accounting and the blending-scalar step.
int hf_filter_deliver(hf_filter* f, hf_ctx* ctx, const void* host_in, void* const* host_out, int max_out, int* n_out, int32_t* kinds) {
    if (!f || !ctx || !host_in || !host_out || !n_out) return HF_ERR_INVALID_ARGUMENT;
    *n_out = 0;
    const int n = hf_filter_begin_source_frame(f);                            // :944-948 (a pure function of the state: calling it twice is harmless)
    if (n > max_out) { *n_out = n; return HF_ERR_INVALID_ARGUMENT; }          // nothing else has happened yet; *n_out = buffers needed
    hf_params p{};
    hf_stats st{};
    int rc = hf_get_params(ctx, &p);
    if (rc) return rc;
    if (f->cfg.auto_adjust) {                                                  // :951 (AUTO_SEARCH_RADIUS_ADJUST)
        if ((rc = hf_get_stats(ctx, &st))) return rc;
        hf_filter_auto_adjust(f, st.ofc_calc_time, &p.search_radius);
        if ((rc = hf_set_params(ctx, &p))) return rc;
    } else {
        f->total_warp_duration = 0.0;
    }
    if ((rc = hf_update_frame(ctx, host_in))) return rc;                       // :953
    if ((rc = hf_get_params(ctx, &p))) return rc;
    if (f->active && p.frame_count >= 3) {                                     // :955-972
        if ((rc = hf_calculate_optical_flow(ctx))) return rc;
        if ((rc = hf_get_stats(ctx, &st))) return rc;
        hf_filter_push_frame_delta(f, p.frame_count, st.total_frame_delta);
    }
    for (int i = 0; i < n; i++) {
        const int scene_change = hf_filter_detect_scene_change(f, p.frame_count);   // :1126-1176
        if (f->active && p.frame_count >= 3 && !scene_change) {                // :1179-1183
            if ((rc = hf_warp_frames(ctx, (float)f->blending_scalar, f->cfg.frame_output_mode))) return rc;
            if (kinds) kinds[i] = 1;
        } else {
            if ((rc = hf_copy_frame(ctx))) return rc;
            if (kinds) kinds[i] = 0;
        }
        if ((rc = hf_download_frame(ctx, host_out[i]))) return rc;             // :1186
        if ((rc = hf_get_stats(ctx, &st))) return rc;
        hf_filter_add_warp_duration(f, st.warp_calc_time);                     // :1189
        hf_filter_advance_blending_scalar(f);                                  // :1192-1197
        (*n_out)++;
    }
    return HF_OK;
}

}  // extern "C"
