// tests/cpp/filter_sanitize.cpp -- the host-only code of the product (hopperrender_amd/csrc/hf_filter.cpp: the caller protocol;
// csrc/hf_hostio.cpp: timeline planner + host-I/O driver) compiled WITH -fsanitize=address,undefined and driven through long random sessions: histories that
// grow and are evicted, seeks, rate changes, degenerate configurations.  `make -C oracle sanitize` builds and runs it.
#include <cstdio>
#include <cstdlib>

#include "hopperflow.h"

// hf_filter.cpp / hf_hostio.cpp are host-only; the HIP half of the library is replaced here by a FAKE calculator that keeps the
// call protocol's observable state (frame count, download counter, one output buffer) in host memory, so that hf_hostio_run --
// ring indexing, drain order, callbacks, filter decisions -- runs end to end under the sanitizers.  (hf_filter_deliver is not
// exercised.)
#include <cstring>
#include <vector>
namespace {
struct FakeCalc { hf_params p{}; uint64_t downloads = 0; uint32_t delta = 0; unsigned char out[64] = {}; int updates = 0; } g_fake;
}
extern "C" {
struct hf_ctx { int unused; };
int hf_get_params(const hf_ctx*, hf_params* p) { *p = g_fake.p; return HF_OK; }
int hf_set_params(hf_ctx*, const hf_params* p) { g_fake.p = *p; return HF_OK; }
int hf_get_stats(hf_ctx*, hf_stats* s) {
    std::memset(s, 0, sizeof(*s));
    s->input_frame_bytes = s->output_frame_bytes = sizeof(g_fake.out);
    s->total_frame_delta = g_fake.delta;
    s->frame_count = g_fake.p.frame_count;
    return HF_OK;
}
int hf_update_frame(hf_ctx*, const void*) { return HF_ERR_STATE; }
int hf_update_frame_async(hf_ctx*, const void* frame) {
    g_fake.p.frame_count++; g_fake.updates++;
    g_fake.delta = ((const unsigned char*)frame)[0] > 200 ? 50000u : 700u + ((const unsigned char*)frame)[1];   // "scene cut" frames
    return HF_OK;
}
int hf_calculate_optical_flow(hf_ctx*) { return HF_OK; }
int hf_wait_flow(hf_ctx*) { return HF_OK; }
int hf_sync(hf_ctx*) { return HF_OK; }
int hf_warp_frames(hf_ctx*, float t, int) { std::memset(g_fake.out, 1 + (int)(t * 100.0f) % 100, sizeof(g_fake.out)); return HF_OK; }
int hf_copy_frame(hf_ctx*) { std::memset(g_fake.out, 0xC0, sizeof(g_fake.out)); return HF_OK; }
int hf_download_frame(hf_ctx*, void*) { return HF_ERR_STATE; }
int hf_download_frame_async(hf_ctx*, void* dst) { std::memcpy(dst, g_fake.out, sizeof(g_fake.out)); g_fake.downloads++; return HF_OK; }
uint64_t hf_downloads_issued(const hf_ctx*) { return g_fake.downloads; }
int hf_wait_download(hf_ctx*, uint64_t i) { return i < g_fake.downloads ? HF_OK : HF_ERR_INVALID_ARGUMENT; }
int hf_host_malloc_pinned(size_t n, void** p) { *p = std::malloc(n); return *p ? HF_OK : HF_ERR_OUT_OF_MEMORY; }
int hf_host_free_pinned(void* p) { std::free(p); return HF_OK; }
const char* hf_last_error(const hf_ctx*) { return ""; }
}

// hf_shard_timeline for every rank of a few (frames, world) combinations + hf_hostio_run of every chunk on the fake calculator:
// outputs must arrive strictly in order, exactly n_outputs of them, kinds consistent with the frames' contents.
static int hostio_session(unsigned long long& checksum) {
    struct User { int64_t next = 0; int64_t first_frame = 0, n_frames = 0; unsigned long long sum = 0; int bad = 0; };
    const int64_t cases[][2] = {{40, 3}, {7, 4}, {64, 1}, {3, 5}, {0, 2}, {25, 8}};
    for (auto& cs : cases) {
        int64_t tiled = 0;
        for (int rank = 0; rank < (int)cs[1]; rank++) {
            hf_timeline_chunk ch{};
            if (hf_shard_timeline(cs[0], (int)cs[1], rank, 417083, 83333 + 1000 * rank % 7, 3, 12, &ch, nullptr, nullptr, 0) != HF_OK) return 10;
            std::vector<int32_t> n_out((size_t)ch.n_periods + 1);
            std::vector<float> t((size_t)ch.n_outputs + 1);
            if (hf_shard_timeline(cs[0], (int)cs[1], rank, 417083, 83333 + 1000 * rank % 7, 3, 12, &ch, n_out.data(), t.data(), ch.n_outputs) != HF_OK) return 11;
            const hf_timeline_chunk before = ch;             // a failed call (scalar buffer too small) must leave *out alone
            if (ch.n_outputs > 0 && (hf_shard_timeline(cs[0], (int)cs[1], rank, 417083, 83333 + 1000 * rank % 7, 3, 12, &ch, n_out.data(), t.data(), ch.n_outputs - 1) == HF_OK ||
                                     std::memcmp(&before, &ch, sizeof(ch)) != 0)) return 12;
            tiled += ch.n_outputs;
            hf_ctx fake{};
            g_fake = FakeCalc{};
            hf_hostio_config hc{};
            hc.struct_size = sizeof(hc);
            hc.in_ring = 3; hc.out_ring = 2 + rank % 3; hc.frame_output_mode = 2; hc.scene_change_threshold = 150;
            hc.source_frame_time = 417083; hc.target_frame_time = 83333;
            hf_hostio* io = nullptr;
            if (hf_hostio_create(&fake, &hc, &io) != HF_OK) return 13;
            User u; u.first_frame = ch.first_frame; u.n_frames = ch.n_frames;
            auto fill = [](void* user, int64_t k, void* dst) -> int {
                User* w = (User*)user;
                if (k < w->first_frame || k >= w->first_frame + w->n_frames) w->bad++;
                std::memset(dst, k == 20 ? 250 : (int)(k % 100), 64);
                return 0;
            };
            auto sink = [](void* user, int64_t i, const void* frame, int32_t kind) -> int {
                User* w = (User*)user;
                if (i != w->next++) w->bad++;
                const unsigned char* f = (const unsigned char*)frame;
                if ((kind == 0) != (f[0] == 0xC0) || f[0] != f[63]) w->bad++;
                w->sum += f[0];
                return 0;
            };
            std::vector<int32_t> kinds((size_t)ch.n_outputs + 1, -1);
            if (hf_hostio_run(io, &ch, n_out.data(), t.data(), fill, sink, &u, kinds.data()) != HF_OK) return 14;
            if (u.bad || u.next != ch.n_outputs || g_fake.updates != ch.n_frames) return 15;
            for (int64_t i = 0; i < ch.n_outputs; i++) if (kinds[(size_t)i] != 0 && kinds[(size_t)i] != 1) return 16;
            uint64_t bi = 0, bo = 0;
            hf_hostio_get_traffic(io, &bi, &bo);
            if (bi != 64ull * (uint64_t)ch.n_frames || bo != 64ull * (uint64_t)ch.n_outputs) return 17;
            // a failing callback stops the run with an error and leaves everything destructible
            auto bad_sink = [](void*, int64_t, const void*, int32_t) -> int { return 7; };
            if (ch.n_outputs > 0 && hf_hostio_run(io, &ch, n_out.data(), t.data(), fill, bad_sink, &u, nullptr) == HF_OK) return 18;
            hf_hostio_destroy(io);
            checksum += u.sum;
        }
        hf_timeline_chunk all{};
        if (hf_shard_timeline(cs[0], 1, 0, 417083, 83333, 3, 12, &all, nullptr, nullptr, 0) != HF_OK) return 19;
        (void)tiled;
    }
    if (hf_shard_timeline(10, 2, 2, 1, 1, 3, 12, nullptr, nullptr, nullptr, 0) != HF_ERR_INVALID_ARGUMENT) return 20;
    if (hf_hostio_create(nullptr, nullptr, nullptr) != HF_ERR_INVALID_ARGUMENT) return 21;
    return 0;
}

int main() {
    unsigned s = 99u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    unsigned long long checksum = 0;
    for (int session = 0; session < 40; session++) {
        hf_filter_config c{};
        c.struct_size = sizeof(c);
        c.scene_change_threshold = session % 5 == 0 ? -1 : (int)(rnd() % 2000);
        c.source_frame_time = session % 7 == 0 ? 0 : 100000 + rnd() % 900000;
        c.target_frame_time = session % 3 == 0 ? 0 : 40000 + rnd() % 400000;
        c.frame_output_mode = (int)(rnd() % 7);
        c.auto_adjust = 1;
        c.active = session % 11 != 0;
        hf_filter* f = nullptr;
        if (hf_filter_create(&c, &f) != HF_OK) return 2;
        int32_t radius = 5;
        uint32_t frame = 0;
        for (int k = 0; k < 3000; k++) {
            if (rnd() % 400 == 0) { hf_filter_new_segment(f, 0.25 + (rnd() % 16) * 0.25); frame = 0; }
            const int n = hf_filter_begin_source_frame(f);
            hf_filter_auto_adjust(f, (rnd() % 1000) * 1e-5, &radius);
            frame++;
            if (frame >= 3) hf_filter_push_frame_delta(f, frame, rnd() % 4000 + (rnd() % 50 == 0 ? 100000u : 0u));
            for (int i = 0; i < n && i < 64; i++) {
                checksum += (unsigned)hf_filter_detect_scene_change(f, frame);
                hf_filter_add_warp_duration(f, (rnd() % 100) * 1e-5);
                hf_filter_advance_blending_scalar(f);
            }
            checksum += (unsigned)radius;
        }
        hf_filter_state st{};
        hf_filter_get_state(f, &st);
        checksum += st.frame_delta_history + st.peak_scene_change_delta;
        hf_filter_destroy(f);
    }
    // argument errors
    if (hf_filter_create(nullptr, nullptr) != HF_ERR_INVALID_ARGUMENT) return 3;
    hf_filter_config bad{};
    hf_filter* f = nullptr;
    if (hf_filter_create(&bad, &f) != HF_ERR_INVALID_ARGUMENT) return 4;
    if (const int rc = hostio_session(checksum)) { std::printf("hostio session failed: %d\n", rc); return rc; }
    std::printf("filter_sanitize ok, checksum %llu\n", checksum);
    return 0;
}
