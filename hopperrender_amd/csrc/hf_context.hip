// hopperrender_amd/csrc/hf_context.hip -- context lifetime and state of the C ABI (include/hopperflow.h): device selection and hf_create /
// hf_destroy (reference: detectDevices + constructor, opticalFlowCalc.cpp:48-109, opticalFlowCalcSDR.cpp:206-325), the public fields
// (hf_get/set_params, hf_get_stats), profiling spans, parity taps and the plain device-memory helpers.  Layout of the ABI: hf_ctx.h.

#include "hf_ctx.h"

using namespace hfi;

namespace {
thread_local std::string g_create_error;
}  // namespace

namespace hfi {
std::shared_mutex g_capture_mutex;
}  // namespace hfi

namespace hfi {

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

hipEvent_t pool_event(hf_ctx* c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// Opens a profiled span on the stream; returns the index of the span or -1.
int span_begin(hf_ctx* c, int kind, hipStream_t stream) {
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

int sync_ctx(hf_ctx* c) {
    if (int rc = leave_warp_stream(c)) return rc;
    HF_HIP(c, hipStreamSynchronize(c->stream));
    if (c->io_in) { HF_HIP(c, hipStreamSynchronize(c->io_in)); HF_HIP(c, hipStreamSynchronize(c->io_out)); }
    collect_spans(c);
    if (c->delta_pending) { c->total_frame_delta = *c->h_total_delta; c->delta_pending = false; }
    finish_flow_timing(c);
    return HF_OK;
}

// hf_memcpy_* have no context, hence no stream of their own, and an extra stream would occupy one of the few hardware
// queues the pair streams need (DESIGN.md "Hardware queues": 50 k -> 42 k frames/s with one more stream alive, 35 k
// with short-lived ones).  They use the legacy stream -- but never while a thread of this process captures a graph: a
// synchronous hipMemcpy there invalidates the capture (HIP error 906, a rare failure of the threads test; 130 errors
// under tools/stress_threads.py).  Captures hold g_capture_mutex shared, these copies exclusively.
int util_copy(int device_index, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (hipSetDevice(device_index) != hipSuccess) return HF_ERR_NO_DEVICE;
    std::unique_lock<std::shared_mutex> lock(g_capture_mutex);
    return hipMemcpy(dst, src, bytes, kind) == hipSuccess ? HF_OK : HF_ERR_HIP;
}

}  // namespace hfi

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
    // SAD tables of the chain (hf_flow.hip): 16 candidates x u16 per 2x2 grid block and axis
    c->sad_nbx = (g.lw + 1) / 2; c->sad_nby = (g.lh + 1) / 2;
    c->sadtab_bytes = (cfg->flags & HF_FLAG_NO_SAD_REUSE) ? 0 : 2 * (size_t)c->sad_nbx * c->sad_nby * 32;
    int rc = HF_OK;
    auto bail = [&](int code) { std::string e = c->err; hf_destroy(c); g_create_error = e; return code; };
    {   // detectDevices (opticalFlowCalc.cpp:45-109): the device must offer the memory, LDS and workgroup size the calculator
        // needs.  The reference prices 9 H S_in + 3 H S_out (HDR worst case) + offset / sum arrays against the device's TOTAL
        // memory and takes the FIRST device that qualifies; this build keeps 3 frames + 3 phase planes + 1 output frame + small
        // tables, priced exactly, and additionally requires that much memory to be FREE on the device it settles on.
        // device_index >= 0 pins the ordinal (one process per GPU); -1 scans like the reference.
        const uint64_t required = 3 * c->in_bytes + 3 * c->pl.bytes + c->out_bytes + c->tables_bytes + c->sums_bytes + c->sadtab_bytes +
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
    if (c->sadtab_bytes) {
        HF_TRY(hipMalloc((void**)&c->sadtab, c->sadtab_bytes));
        HF_TRY(hipMalloc((void**)&c->still_count, 64));
        HF_TRY(hipMemsetAsync(c->still_count, 0, 64, c->stream));
    }   // (every entry is written before it is read inside a chain)
    HF_TRY(hipMalloc((void**)&c->sums, c->sums_bytes));
    HF_TRY(hipMemsetAsync(c->sums, 0, c->sums_bytes, c->stream));
    HF_TRY(hipMalloc((void**)&c->d_probe, 64 * sizeof(float)));
    // m_totalFrameDelta is stored by the chain straight into mapped pinned memory (reference: blocking 4-byte
    // readback in the middle of the chain, opticalFlowCalcSDR.cpp:91-94)
    HF_TRY(hipHostMalloc((void**)&c->h_total_delta, 64, hipHostMallocMapped));
    *c->h_total_delta = 0;
    c->h_total_delta[1] = 0xFFFFFFFFu;                  // (no chain has published its content hint yet)
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
    if (c->sadtab) hipFree(c->sadtab);
    if (c->still_count) hipFree(c->still_count);
    if (c->counters) hipFree(c->counters);
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

int hf_sync(hf_ctx* c) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    return sync_ctx(c);
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
    out->sad_tables = c->sadtab && (c->batch ? c->batch->members[0]->tab_mode : c->tab_mode) ? 1 : 0;
    out->still_share = c->still_share;
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

int hf_debug_counters_enable(hf_ctx* c, int on) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    if (int rc = sync_ctx(c)) return rc;
    // the pointer is part of every captured launch: cached graphs of this context and of its batch are stale now
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    if (c->batch) {
        for (auto& kv : c->batch->graphs) hipGraphExecDestroy(kv.second);
        c->batch->graphs.clear();
    }
    if (on) {
        if (!c->counters) HF_HIP(c, hipMalloc((void**)&c->counters, hf::kCounterWords * sizeof(uint32_t)));
        HF_HIP(c, hipMemsetAsync(c->counters, 0, hf::kCounterWords * sizeof(uint32_t), c->stream));
        return sync_ctx(c);
    }
    if (c->counters) { hipFree(c->counters); c->counters = nullptr; }
    return HF_OK;
}

int hf_debug_counters_read(hf_ctx* c, hf_debug_counters* out, int reset) {
    HF_CHECK_CTX(c);
    if (!out) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_debug_counters_read: null");
    if (!c->counters) return fail(c, HF_ERR_STATE, "hf_debug_counters_read: hf_debug_counters_enable(ctx, 1) first");
    if (int rc = set_device(c)) return rc;
    if (int rc = sync_ctx(c)) return rc;
    uint32_t w[hf::kCounterWords];
    HF_HIP(c, hipMemcpy(w, c->counters, sizeof(w), hipMemcpyDeviceToHost));
    if (reset) HF_HIP(c, hipMemset(c->counters, 0, sizeof(w)));
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < 3; i++) out->warp_workgroups[i] = w[hf::kCounterWarp + i];
    for (int k = 0; k < 16; k++) {
        for (int ax = 0; ax < 2; ax++) {
            out->level_windows[k][ax] = w[hf::kCounterLevels + 4 * k + 2 * ax];
            out->level_reused[k][ax] = w[hf::kCounterLevels + 4 * k + 2 * ax + 1];
        }
        out->level_window_size[k] = k < (int)c->levels.size() && k < c->last_iterations ? c->levels[(size_t)k].window : 0;
    }
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

// ---- shader clock under load ----
// One wave reads the shader-cycle counter (s_memtime) and the 100 MHz reference counter (s_memrealtime) around a spin of `ticks` reference
// ticks: cycles / time = the clock the shader array actually runs at while everything else the process has queued keeps the device
// busy (MI355X_MICROARCH.md "DVFS give-back": the chip lowers its clock under load, by a device-dependent amount -- what makes the boxes
// of a pool differ).  Reads no buffer and writes 16 bytes; in a translation unit of its own so that the product kernels' code is untouched.
__global__ void clock_probe_kernel(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

int hf_clock_probe(int device_index, int duration_us, double* shader_mhz) {
    if (!shader_mhz || duration_us < 10 || duration_us > 100000) return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_clock_probe: bad argument");
    *shader_mhz = 0.0;
    if (hipSetDevice(device_index) != hipSuccess) return fail(nullptr, HF_ERR_NO_DEVICE, "hf_clock_probe: bad device %d", device_index);
    unsigned long long* res = nullptr;
    hipStream_t s = nullptr;
    if (hipHostMalloc((void**)&res, 2 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return fail(nullptr, HF_ERR_OUT_OF_MEMORY, "hf_clock_probe: hipHostMalloc failed");
    res[0] = res[1] = 0;
    int rc = HF_OK;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) rc = fail(nullptr, HF_ERR_HIP, "hf_clock_probe: cannot create a stream");
    if (rc == HF_OK) {
        clock_probe_kernel<<<1, 64, 0, s>>>((unsigned long long)duration_us * 100ull, res);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = fail(nullptr, HF_ERR_HIP, "hf_clock_probe: launch failed");
    }
    if (rc == HF_OK && res[1] > 0) *shader_mhz = (double)res[0] / ((double)res[1] / 100.0);
    if (s) hipStreamDestroy(s);
    hipHostFree(res);
    return rc;
}

// ---- what this device's HBM sustains for a plain streaming copy (the yardstick a bandwidth-bound pipeline should be held against) ----
// 16 bytes per lane, 4 loads in flight per thread, non-temporal loads and stores: the access shape of the guide's "float4 copy".
__global__ __launch_bounds__(256) void hbm_copy_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const v4 a = __builtin_nontemporal_load((const v4*)src + i), b = __builtin_nontemporal_load((const v4*)src + i + stride);
        const v4 c = __builtin_nontemporal_load((const v4*)src + i + 2 * stride), d = __builtin_nontemporal_load((const v4*)src + i + 3 * stride);
        __builtin_nontemporal_store(a, (v4*)dst + i); __builtin_nontemporal_store(b, (v4*)dst + i + stride);
        __builtin_nontemporal_store(c, (v4*)dst + i + 2 * stride); __builtin_nontemporal_store(d, (v4*)dst + i + 3 * stride);
    }
    for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load((const v4*)src + i), (v4*)dst + i);
}

int hf_hbm_copy_probe(int device_index, size_t bytes, int repeats, double* read_plus_write_GBps) {
    if (!read_plus_write_GBps || bytes < (1u << 20) || repeats < 1 || repeats > 100) return fail(nullptr, HF_ERR_INVALID_ARGUMENT, "hf_hbm_copy_probe: bad argument");
    *read_plus_write_GBps = 0.0;
    if (hipSetDevice(device_index) != hipSuccess) return fail(nullptr, HF_ERR_NO_DEVICE, "hf_hbm_copy_probe: bad device %d", device_index);
    bytes &= ~(size_t)15;
    void *a = nullptr, *b = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = HF_OK;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) rc = fail(nullptr, HF_ERR_OUT_OF_MEMORY, "hf_hbm_copy_probe: hipMalloc failed");
    if (rc == HF_OK && (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess))
        rc = fail(nullptr, HF_ERR_HIP, "hf_hbm_copy_probe: stream / event creation failed");
    if (rc == HF_OK) {
        (void)hipMemsetAsync(a, 0x5A, bytes, s);
        const size_t n16 = bytes / 16;
        // one pass of four 16-byte elements per thread, no loop: measured fastest on MI355X (tools/ubench/copy_rate.hip, 2 GiB: 5.7 TB/s
        // with 131,072 workgroups against 4.8-5.3 TB/s with 2,048-32,768 looping ones; plain or non-temporal, 1-8 loads in flight: +-3 %)
        const unsigned blocks = (unsigned)((n16 + 1023) / 1024);
        float best = 0.f;
        for (int r = 0; r <= repeats && rc == HF_OK; r++) {      // (the first pass warms up)
            (void)hipEventRecord(e0, s);
            hbm_copy_probe_kernel<<<blocks, 256, 0, s>>>((const uint4*)a, (uint4*)b, n16);
            (void)hipEventRecord(e1, s);
            float ms = 0.f;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(nullptr, HF_ERR_HIP, "hf_hbm_copy_probe: launch failed");
            else if (r > 0 && ms > 0.f) { const float g = (float)(2.0 * (double)bytes / (ms * 1e-3) / 1e9); best = g > best ? g : best; }
        }
        *read_plus_write_GBps = (double)best;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (s) hipStreamDestroy(s);
    if (a) hipFree(a);
    if (b) hipFree(b);
    return rc;
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

int hf_memcpy_h2d(int device_index, void* d, const void* h, size_t bytes) { return util_copy(device_index, d, h, bytes, hipMemcpyHostToDevice); }

int hf_memcpy_d2h(int device_index, void* h, const void* d, size_t bytes) { return util_copy(device_index, h, d, bytes, hipMemcpyDeviceToHost); }

}  // extern "C"
