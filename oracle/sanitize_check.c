/* oracle/sanitize_check.c -- TEST INFRASTRUCTURE ONLY.  Runs every entry point of the oracle (hf_oracle.c) on seeded
 * frames of awkward shapes under -fsanitize=address,undefined (`make -C oracle sanitize`): ragged sizes, strides larger
 * than the width, 4-row frames, radius 2..16, every output mode, both element types.  SURVEY.md section 5 lists
 * sanitizer builds of the CPU side as an auxiliary (reference toggle: common/platform.props:22); GPU ASan is not
 * available on the pool, so this covers the checker the GPU results are pinned to.  Exit code 0 = clean. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hf_oracle.h"

static uint32_t rng_state = 12345u;
static uint32_t rng(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static void* make_frame(const hfo_geom* g, size_t* bytes) {
    const size_t n = ((size_t)g->H + g->H / 2) * g->in_stride, es = g->hdr ? 2 : 1;
    unsigned char* f = malloc(n * es);   /* exact size: an out-of-bounds read is an ASan report */
    for (size_t i = 0; i < n * es; i++) f[i] = (unsigned char)rng();
    *bytes = n * es;
    return f;
}

int main(void) {
    static const int shapes[][5] = {   /* hdr, H, W, in_stride, out_stride */
        {0, 180, 320, 0, 0}, {1, 180, 320, 0, 0}, {0, 94, 166, 176, 192}, {1, 66, 118, 128, 120}, {0, 4, 4, 0, 0},
        {1, 6, 8, 0, 0}, {0, 272, 482, 0, 0}, {1, 360, 640, 0, 0},
    };
    unsigned long long checksum = 0;
    for (unsigned s = 0; s < sizeof(shapes) / sizeof(shapes[0]); s++) {
        hfo_geom g;
        hfo_make_geom(&g, shapes[s][0], shapes[s][1], shapes[s][2], shapes[s][3], shapes[s][4], s == 6 ? 1000 : 270);
        size_t fb;
        void* f0 = make_frame(&g, &fb);
        void* f1 = make_frame(&g, &fb);
        void* f2 = make_frame(&g, &fb);
        const size_t N = (size_t)g.lw * g.lh, es = g.hdr ? 2 : 1;
        const size_t out_bytes = ((size_t)g.H + g.H / 2) * g.out_stride * es;
        int16_t* off = malloc(2 * N * sizeof(int16_t));
        int16_t* blur = malloc(2 * N * sizeof(int16_t));
        void* out = malloc(out_bytes);
        static const int radii[] = {2, 5, 9, 16};
        for (unsigned r = 0; r < 4; r++) {
            uint32_t total = 0;
            hfo_diag d = {0};
            hfo_calculate_optical_flow(f1, f2, &g, radii[r], r == 1 ? 2 : 0, 8 - (int)r, 6 + (int)r, r == 3 ? 16 : 4, off, blur, &total, &d);
            checksum += total + d.oob_samples;
            for (int mode = 0; mode <= 6; mode++) {
                memset(out, 0, out_bytes);
                hfo_warp_frames(f0, f1, blur, out, &g, 0.3996f, mode, 0.0f, 255.0f);
                checksum += ((unsigned char*)out)[out_bytes / 2];
            }
            hfo_warp_frames(f0, f1, blur, out, &g, 1.0f, 2, 16.0f, 235.0f);
            hfo_copy_frame(f2, out, &g, 16.0f, 235.0f);
            checksum += ((unsigned char*)out)[out_bytes - 1];
        }
        free(f0); free(f1); free(f2); free(off); free(blur); free(out);
    }
    printf("sanitize_check ok, checksum %llu\n", checksum);
    return 0;
}
