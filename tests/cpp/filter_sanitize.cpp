// tests/cpp/filter_sanitize.cpp -- the host-only protocol code (hopperrender_amd/csrc/hf_filter.cpp: no HIP calls outside
// hf_filter_deliver) compiled WITH -fsanitize=address,undefined and driven through long random sessions: histories that
// grow and are evicted, seeks, rate changes, degenerate configurations.  `make -C oracle sanitize` builds and runs it.
#include <cstdio>
#include <cstdlib>

#include "hopperflow.h"

// hf_filter_deliver is the only function of hf_filter.cpp that calls into the HIP half of the library; it is not
// exercised here, the stubs only satisfy the linker.
extern "C" {
int hf_get_params(const hf_ctx*, hf_params*) { return HF_ERR_STATE; }
int hf_set_params(hf_ctx*, const hf_params*) { return HF_ERR_STATE; }
int hf_get_stats(hf_ctx*, hf_stats*) { return HF_ERR_STATE; }
int hf_update_frame(hf_ctx*, const void*) { return HF_ERR_STATE; }
int hf_calculate_optical_flow(hf_ctx*) { return HF_ERR_STATE; }
int hf_warp_frames(hf_ctx*, float, int) { return HF_ERR_STATE; }
int hf_copy_frame(hf_ctx*) { return HF_ERR_STATE; }
int hf_download_frame(hf_ctx*, void*) { return HF_ERR_STATE; }
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
    std::printf("filter_sanitize ok, checksum %llu\n", checksum);
    return 0;
}
