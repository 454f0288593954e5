/* examples/hf_batch_driver.c -- a throughput driver written against the C ABI only (include/hopperflow.h):
 * two independent NV12 clips, one hf_batch, device-resident frames: per source period ONE phase-plane launch, ONE batched
 * refinement chain and ONE fused warp launch for both clips.
 *
 *   gcc -std=c11 -Iinclude examples/hf_batch_driver.c -Lhopperrender_amd/lib -lhopperflow -Wl,-rpath,$PWD/hopperrender_amd/lib -o hf_batch_driver
 *   ./hf_batch_driver [periods]
 *
 * Prints one FNV-1a checksum per output frame (tests/test_batch_gpu.py compares them with the Python path). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hopperflow.h"

enum { H = 180, W = 320, CLIPS = 2, NOUT = 3 };
#define FRAME_BYTES ((size_t)W * H * 3 / 2)

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != HF_OK) { fprintf(stderr, "%s failed: %d\n", #call, rc_); exit(1); }   \
    } while (0)

/* synthetic clip: a moving diagonal texture, different per clip (tests regenerate the same bytes in numpy) */
static void make_frame(uint8_t* f, int clip, int k) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            f[(size_t)y * W + x] = (uint8_t)(((x + 3 * k + 5 * clip) * 7 + (y + 2 * k) * 13 + (((x + 3 * k) >> 4) ^ ((y + 2 * k) >> 4)) * 29) & 0xFF);
    for (int y = 0; y < H / 2; y++)
        for (int x = 0; x < W; x++)
            f[(size_t)H * W + (size_t)y * W + x] = (uint8_t)(128 + (((x >> 1) + k + clip) * 3 + y * 5) % 64 - 32);
}

static uint64_t fnv1a(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    const int periods = argc > 1 ? atoi(argv[1]) : 5;
    hf_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.frame_height = H; cfg.frame_width = W;
    cfg.delta_scalar = 8; cfg.neighbor_scalar = 6;
    cfg.black_level = 0.0f; cfg.white_level = 255.0f;
    cfg.max_calc_res = 270;
    cfg.search_radius = 12;
    cfg.flags = HF_FLAG_ASYNC | HF_FLAG_NO_TIMING;

    hf_ctx* ctx[CLIPS];
    for (int c = 0; c < CLIPS; c++) CHECK(hf_create(&cfg, &ctx[c]));
    hf_batch* batch = NULL;
    CHECK(hf_batch_create(ctx, CLIPS, &batch));

    /* device-resident source frames (a decoder would own these) and output buffers */
    const int n_frames = periods + 2;
    uint8_t* host = (uint8_t*)malloc(FRAME_BYTES);
    void** src = (void**)malloc(sizeof(void*) * CLIPS * n_frames);
    for (int c = 0; c < CLIPS; c++)
        for (int k = 0; k < n_frames; k++) {
            CHECK(hf_device_malloc(0, FRAME_BYTES, &src[c * n_frames + k]));
            make_frame(host, c, k);
            CHECK(hf_memcpy_h2d(0, src[c * n_frames + k], host, FRAME_BYTES));
        }
    void* out[CLIPS][NOUT];
    for (int c = 0; c < CLIPS; c++)
        for (int i = 0; i < NOUT; i++) CHECK(hf_device_malloc(0, FRAME_BYTES, &out[c][i]));

    const float t[NOUT] = {0.0f, 0.3996f, 0.7992f};
    for (int c = 0; c < CLIPS; c++)                       /* prime the ring with the first two frames */
        for (int k = 0; k < 2; k++) CHECK(hf_update_frame_device_ref(ctx[c], src[c * n_frames + k]));

    int n_out[CLIPS];
    float ts[CLIPS][HF_MAX_PERIOD_OUTPUTS];
    void* outs[CLIPS][HF_MAX_PERIOD_OUTPUTS];
    memset(outs, 0, sizeof(outs));
    for (int c = 0; c < CLIPS; c++) {
        n_out[c] = NOUT;
        for (int i = 0; i < NOUT; i++) { ts[c][i] = t[i]; outs[c][i] = out[c][i]; }
    }
    for (int p = 0; p < periods; p++) {
        const void* next[CLIPS];
        for (int c = 0; c < CLIPS; c++) next[c] = src[c * n_frames + p + 2];
        /* the whole source period of both clips in ONE call: updateFrame of both (one phase-plane launch), both flow
         * calculations in one set of launches, all outputs of both clips in one fused warp launch */
        CHECK(hf_batch_run_period(batch, next, 1, n_out, &ts[0][0], (void* const*)&outs[0][0], 2));
        for (int c = 0; c < CLIPS; c++) {
            CHECK(hf_sync(ctx[c]));
            hf_stats st;
            CHECK(hf_get_stats(ctx[c], &st));
            for (int i = 0; i < NOUT; i++) {
                CHECK(hf_memcpy_d2h(0, host, out[c][i], FRAME_BYTES));
                printf("period %d clip %d out %d delta %u fnv %016llx\n", p, c, i, st.total_frame_delta, (unsigned long long)fnv1a(host, FRAME_BYTES));
            }
        }
    }

    hf_batch_destroy(batch);                               /* before its members */
    for (int c = 0; c < CLIPS; c++) hf_destroy(ctx[c]);
    for (int i = 0; i < CLIPS * n_frames; i++) hf_device_free(0, src[i]);
    for (int c = 0; c < CLIPS; c++)
        for (int i = 0; i < NOUT; i++) hf_device_free(0, out[c][i]);
    free(src); free(host);
    return 0;
}
