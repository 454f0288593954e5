/* oracle/hf_oracle.c -- TEST INFRASTRUCTURE ONLY (see hf_oracle.h).
 *
 * Scalar C restatement of the reference hot path, written from the cited lines; no reference
 * text is copied.  Build with -ffp-contract=off (oracle/Makefile) so the fp32 expressions are
 * evaluated operation by operation.
 */
#include "hf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Default = flavour 1/1: bit-exact with the reference as it actually runs on an MI355X through
 * AMD OpenCL (the tests/golden fixtures); flavour 0/0 is the strict-IEEE reading of the same source. */
static int g_blend_flavour = 1;
static int g_levels_flavour = 1;
void hfo_set_blend_flavour(int f) { g_blend_flavour = f; }
void hfo_set_levels_flavour(int f) { g_levels_flavour = f; }

/* ---- geometry / schedule -------------------------------------------------------------- */

void hfo_make_geom(hfo_geom* g, int hdr, int H, int W, int in_stride, int out_stride, int max_calc_res) {
    g->hdr = hdr ? 1 : 0;
    g->H = H;
    g->W = W;
    g->in_stride = in_stride > 0 ? in_stride : W;    /* opticalFlowCalcSDR.cpp:212 */
    g->out_stride = out_stride > 0 ? out_stride : W; /* :213 */
    int rs = 0;
    while ((H >> rs) > max_calc_res) rs++;           /* :217-220 */
    g->rs = rs;
    /* :221-222  ceil(dim / 2^rs) in double; exact for every int dim */
    g->lw = (int)ceil((double)W / pow(2.0, rs));
    g->lh = (int)ceil((double)H / pow(2.0, rs));
}

int hfo_initial_window(int lw, int lh) {
    /* opticalFlowCalcSDR.cpp:49-59 */
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

int hfo_iterations(int ws0, int requested) {
    /* opticalFlowCalcSDR.cpp:62-65 ; log2 of a power of two, truncated to int */
    int l2 = 0;
    while ((1 << (l2 + 1)) <= ws0) l2++;
    if (ws0 < 1) l2 = 0;
    if (requested == 0 || requested > l2) return l2;
    return requested;
}

int hfo_rel_offset(int layer, int R) {
    /* calcDeltaSumsKernelSDR.h:70-74 ; adjustOffsetArrayKernelSDR.h:16-19 */
    int16_t rel = (int16_t)((layer % R) - (R / 2));
    rel = (int16_t)(rel * rel * (rel > 0 ? 1 : -1));
    return rel;
}

/* ---- element access ------------------------------------------------------------------- */

static inline unsigned top8(const void* f, int hdr, long idx) {
    /* SDR: the byte itself; HDR: sample >> 8 (calcDeltaSumsKernelHDR.h:98-100) */
    return hdr ? (unsigned)(((const uint16_t*)f)[idx] >> 8) : (unsigned)((const uint8_t*)f)[idx];
}
static inline unsigned load_el(const void* f, int hdr, long idx) {
    return hdr ? (unsigned)((const uint16_t*)f)[idx] : (unsigned)((const uint8_t*)f)[idx];
}
static inline void store_el(void* f, int hdr, long idx, unsigned v) {
    if (hdr) ((uint16_t*)f)[idx] = (uint16_t)v; else ((uint8_t*)f)[idx] = (uint8_t)v;
}
static inline unsigned absdiff_u(unsigned a, unsigned b) { return a > b ? a - b : b - a; }
/* OpenCL clamp(x, lo, hi) = min(max(x, lo), hi); the order matters only when lo > hi (mirrorCoordinate on a plane of
 * 2 rows, i.e. 4-row frames: clamp(r, 1, 0)), where the reference as run on the MI355X returns hi -- checked live
 * with tools/debug_tiny_ref.py */
static inline int clampi(int v, int lo, int hi) { const int m = v < lo ? lo : v; return m > hi ? hi : m; }

/* ---- calcDeltaSums -------------------------------------------------------------------- */

void hfo_calc_delta_sums(uint32_t* sums, const void* frame1, const void* frame2, const int16_t* offsets,
                         const hfo_geom* g, int window, int R, int iteration, int step,
                         int delta_scalar, int neighbor_scalar, hfo_diag* diag) {
    const int lw = g->lw, lh = g->lh, W = g->W, H = g->H, S = g->in_stride, hdr = g->hdr;
    const long N = (long)lw * lh;
    for (int cz = 0; cz < R; cz++) {
        const int rel = hfo_rel_offset(cz, R);
        for (int cy = 0; cy < lh; cy++) {
            for (int cx = 0; cx < lw; cx++) {
                const long t2d = (long)cy * lw + cx;
                const int scx = cx << g->rs, scy = cy << g->rs;      /* :50-51 */
                const int16_t ideal_x = offsets[t2d];                 /* :65 */
                const int16_t ideal_y = offsets[N + t2d];             /* :66 */
                /* :69-77 : X searched on even steps, Y on odd steps */
                const int16_t off_x = (int16_t)(ideal_x + ((step & 1) ? 0 : rel));
                const int16_t off_y = (int16_t)(ideal_y + ((step & 1) ? rel : 0));
                int ncx = scx + off_x, ncy = scy + off_y;             /* :78-79 */
                uint32_t delta = 0;
                if (!(scx < 0 || scx >= W || scy < 0 || scy >= H)) {  /* :82 */
                    /* :86-95 single reflection at the frame edge */
                    if (ncx >= W) ncx = W - (ncx - W + 1); else if (ncx < 0) ncx = -ncx - 1;
                    if (ncy >= H) ncy = H - (ncy - H + 1); else if (ncy < 0) ncy = -ncy - 1;
                    if (ncx < 0 || ncx >= W || ncy < 0 || ncy >= H) {
                        /* Reference reads out of bounds here (UB).  The oracle clamps so it stays
                         * defined, and counts the event so tests can exclude such cases. */
                        if (diag) diag->oob_samples++;
                        ncx = clampi(ncx, 0, W - 1);
                        ncy = clampi(ncy, 0, H - 1);
                    }
                    const long uv = (long)H * S;
                    /* :98-100 */
                    delta = absdiff_u(top8(frame1, hdr, (long)ncy * S + ncx), top8(frame2, hdr, (long)scy * S + scx))
                          + absdiff_u(top8(frame1, hdr, uv + (long)(ncy >> 1) * S + (ncx & ~1)),
                                      top8(frame2, hdr, uv + (long)(scy >> 1) * S + (scx & ~1)))
                          + absdiff_u(top8(frame1, hdr, uv + (long)(ncy >> 1) * S + (ncx & ~1) + 1),
                                      top8(frame2, hdr, uv + (long)(scy >> 1) * S + (scx & ~1) + 1));
                    delta <<= delta_scalar;                           /* :101 */
                }
                /* :105-109 */
                const int16_t searched = step ? off_y : off_x;
                const uint32_t offset_bias = (uint32_t)(uint16_t)(searched < 0 ? -searched : searched);
                /* :112-144 */
                uint32_t neighbor_bias = 0;
                if (iteration >= 4) {
                    static const int dirs[4][2] = {{0, 2}, {2, 0}, {-2, 0}, {0, -2}};
                    const int16_t* plane = offsets + (step ? N : 0);
                    for (int i = 0; i < 4; i++) {
                        const int nx = clampi(cx + dirs[i][0] * window, 0, lw - 1);
                        const int ny = clampi(cy + dirs[i][1] * window, 0, lh - 1);
                        const int nb = plane[(long)ny * lw + nx];
                        const int d = nb - searched;
                        neighbor_bias += (uint32_t)(uint16_t)(d < 0 ? -d : d); /* abs_diff(short,short) -> ushort */
                    }
                    neighbor_bias <<= neighbor_scalar;
                }
                const uint32_t cost = delta + offset_bias + neighbor_bias;
                /* :146-190 : plain sum per window, kept at the window-origin slot */
                const int wx = (cx / window) * window, wy = (cy / window) * window;
                sums[(long)cz * N + (long)wy * lw + wx] += cost;
            }
        }
    }
}

/* ---- determineLowestLayer / adjustOffsetArray ------------------------------------------ */

void hfo_determine_lowest_layer(const uint32_t* sums, uint8_t* lowest, int window, int R, int lh, int lw) {
    const long N = (long)lw * lh;
    for (int cy = 0; cy < lh; cy += window)
        for (int cx = 0; cx < lw; cx += window) {
            const long p = (long)cy * lw + cx;
            uint8_t best = 0;
            for (int z = 1; z < R; z++)                               /* :19-24 strict '<' */
                if (sums[z * N + p] < sums[best * N + p]) best = (uint8_t)z;
            lowest[p] = best;
        }
}

void hfo_adjust_offsets(int16_t* offsets, const uint8_t* lowest, int window, int R, int lh, int lw, int step) {
    const long N = (long)lw * lh;
    for (int cy = 0; cy < lh; cy++)
        for (int cx = 0; cx < lw; cx++) {
            const int wx = (cx / window) * window, wy = (cy / window) * window;
            const int rel = hfo_rel_offset(lowest[(long)wy * lw + wx], R);
            int16_t* o = &offsets[(step & 1) * N + (long)cy * lw + cx];
            *o = (int16_t)(*o + rel);
        }
}

/* ---- blurFlow --------------------------------------------------------------------------- */

static inline int mirror_flow(int pos, int dim) {
    /* blurFlowKernelSDR.h:7-14 */
    if (pos >= dim) return dim - (pos - dim + 1);
    if (pos < 0) return -pos - 1;
    return pos;
}

void hfo_blur_flow(const int16_t* offsets, int16_t* blurred, int lh, int lw, int radius) {
    const long N = (long)lw * lh;
    if (radius < 1) { memcpy(blurred, offsets, 2 * N * sizeof(int16_t)); return; } /* :26-29 */
    const int ksize = (2 * radius) * (2 * radius);
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < lh; y++)
            for (int x = 0; x < lw; x++) {
                int sum = 0;
                for (int ky = -radius; ky < radius; ky++)             /* :82-86 taps -r .. r-1 */
                    for (int kx = -radius; kx < radius; kx++) {
                        const int yy = clampi(mirror_flow(y + ky, lh), 0, lh - 1);
                        const int xx = clampi(mirror_flow(x + kx, lw), 0, lw - 1);
                        sum += offsets[z * N + (long)yy * lw + xx];
                    }
                blurred[z * N + (long)y * lw + x] = (int16_t)(sum / ksize); /* :89-90 C truncation */
            }
}

/* ---- calculateOpticalFlow ---------------------------------------------------------------- */

void hfo_calculate_optical_flow(const void* frame1, const void* frame2, const hfo_geom* g, int R,
                                int iterations, int delta_scalar, int neighbor_scalar, int blur_radius,
                                int16_t* offsets_out, int16_t* blurred_out, uint32_t* total_frame_delta,
                                hfo_diag* diag) {
    const long N = (long)g->lw * g->lh;
    uint32_t* sums = (uint32_t*)malloc((size_t)R * N * sizeof(uint32_t));
    uint8_t* lowest = (uint8_t*)calloc((size_t)N, 1);
    int window = hfo_initial_window(g->lw, g->lh);
    const int iters = hfo_iterations(window, iterations);
    memset(offsets_out, 0, 2 * N * sizeof(int16_t));                  /* opticalFlowCalcSDR.cpp:68-69 */
    for (int iter = 0; iter < iters; iter++) {
        for (int step = 0; step < 2; step++) {
            memset(sums, 0, (size_t)R * N * sizeof(uint32_t));        /* :75-76 */
            hfo_calc_delta_sums(sums, frame1, frame2, offsets_out, g, window, R, iter, step,
                                delta_scalar, neighbor_scalar, diag);
            if (iter == 0 && step == 0 && total_frame_delta) {
                /* :91-94 ; HDR divisor 6 (opticalFlowCalcHDR.cpp:93) */
                uint32_t v = sums[(long)((R / 2) - 1) * N];
                v /= (uint32_t)(g->lh * g->lw * (g->hdr ? 6 : 10));
                *total_frame_delta = v;
            }
            hfo_determine_lowest_layer(sums, lowest, window, R, g->lh, g->lw);
            hfo_adjust_offsets(offsets_out, lowest, window, R, g->lh, g->lw, step);
        }
        window = (window >> 1) > 1 ? (window >> 1) : 1;               /* :110 */
    }
    if (blurred_out) hfo_blur_flow(offsets_out, blurred_out, g->lh, g->lw, blur_radius); /* :115-116 */
    free(sums);
    free(lowest);
}

/* ---- levels / warp / copy ----------------------------------------------------------------- */

/* Reciprocal used by levels flavour 1.  The reference's OpenCL build on gfx950 lowers the fp32
 * division to x * v_rcp_f32(y); v_rcp_f32 is accurate to 1 ulp but not always correctly rounded
 * (measured: y = 46080 comes out 1 ulp low), so tests on the GPU box may inject the device's
 * own values through hfo_set_rcp_override(); without an override 1/y is correctly rounded. */
static float g_rcp_in[8], g_rcp_out[8];
static int g_rcp_n = 0;
void hfo_set_rcp_override(int n, const float* y, const float* rcp_y) {
    g_rcp_n = n > 8 ? 8 : (n < 0 ? 0 : n);
    for (int i = 0; i < g_rcp_n; i++) { g_rcp_in[i] = y[i]; g_rcp_out[i] = rcp_y[i]; }
}
static inline float dev_rcp(float y) {
    for (int i = 0; i < g_rcp_n; i++) if (g_rcp_in[i] == y) return g_rcp_out[i];
    volatile float r = 1.0f / y;
    return r;
}

static inline unsigned levels_y(float value, float black, float white, int hdr) {
    /* warpFrameKernelSDR.h:3-5 / HDR :3-5 ; result converted float -> unsigned short (truncation) */
    const float maxv = hdr ? 65535.0f : 255.0f;
    float v;
    if (g_levels_flavour == 1) v = ((value - black) * dev_rcp(white - black)) * maxv;
    else v = (value - black) / (white - black) * maxv;
    v = fmaxf(fminf(v, maxv), 0.0f);
    return (unsigned)(uint16_t)v;
}
static inline unsigned levels_uv(float value, float white, int hdr) {
    /* warpFrameKernelSDR.h:7-9 / HDR :7-9 */
    const float maxv = hdr ? 65535.0f : 255.0f, mid = hdr ? 32768.0f : 128.0f;
    float v;
    if (g_levels_flavour == 1) v = fmaf((value - mid) * dev_rcp(white), maxv, mid);
    else v = (value - mid) / white * maxv + mid;
    v = fmaxf(fminf(v, maxv), 0.0f);
    return (unsigned)(uint16_t)v;
}

static inline int mirror_warp(int pos, int dim) {
    /* warpFrameKernelSDR.h:12-20 */
    int res = pos;
    if (pos >= dim - 1) res = pos - ((pos - (dim - 2)) * 2);
    else if (pos < 1) res = -pos + 1;
    return clampi(res, 1, dim - 2);
}

/* warpFrameKernelSDR.h:23-113 (HDR :23-113) : HSV flow visualisation, diagnostic mode 3 */
static unsigned visualize_flow(int16_t ox, int16_t oy, unsigned curr, int channel, int res_impact, int hdr) {
    unsigned char r = 0, gch = 0, b = 0;
    const unsigned ax = (uint16_t)(ox < 0 ? -ox : ox), ay = (uint16_t)(oy < 0 ? -oy : oy);
    if (!((float)ax < 1.0f && (float)ay < 1.0f)) {
        const float angle_rad = atan2f((float)oy, (float)ox);
        float angle_deg = angle_rad * (180.0f / 3.14159274101257f);
        if (angle_deg < 0) angle_deg += 360.0f;
        angle_deg = fmodf(angle_deg, 360.0f);
        if (angle_deg < 0) angle_deg += 360.0f;
        const float hue = angle_deg / 360.0f;
        const int h_i = (int)(hue * 6.0f);
        const float f = hue * 6.0f - h_i;
        const float q = 1.0f - f;
        switch (h_i % 6) {
            case 0: r = 255; gch = (unsigned char)(f * 255.0f); b = 0; break;
            case 1: r = (unsigned char)(q * 255.0f); gch = 255; b = 0; break;
            case 2: r = 0; gch = 255; b = (unsigned char)(f * 255.0f); break;
            case 3: r = 0; gch = (unsigned char)(q * 255.0f); b = 255; break;
            case 4: r = (unsigned char)(f * 255.0f); gch = 0; b = 255; break;
            case 5: r = 255; gch = 0; b = (unsigned char)(q * 255.0f); break;
            default: r = gch = b = 0; break;
        }
        const int mag = (int)ax + (int)ay;
        r = (unsigned char)fmaxf(fminf((float)r / 255.0f * (float)mag * (float)res_impact, 255.0f), 0.0f);
        gch = (unsigned char)fmaxf(fminf((float)gch / 255.0f * (float)ay * 2.0f * (float)res_impact, 255.0f), 0.0f);
        b = (unsigned char)fmaxf(fminf((float)b / 255.0f * (float)mag * (float)res_impact, 255.0f), 0.0f);
    }
    if (channel == 0) {
        const unsigned y = (unsigned)fmaxf(fminf(r * 0.299f + gch * 0.587f + b * 0.114f, 255.0f), 0.0f);
        if (hdr) return (uint16_t)(((uint16_t)y << 7) + (curr >> 1));
        return (uint8_t)(((uint8_t)y >> 1) + ((uint8_t)curr >> 1));
    }
    float c;
    if (channel == 1) c = fmaxf(fminf(r * -0.168736f + gch * -0.331264f + b * 0.5f + 128.0f, 255.0f), 0.0f);
    else c = fmaxf(fminf(r * 0.5f + gch * -0.418688f + b * -0.081312f + 128.0f, 255.0f), 0.0f);
    if (hdr) return (uint16_t)((uint16_t)c << 8);
    return (uint8_t)c;
}

static void warp_plane(const void* A, const void* B, const int16_t* flow, void* out, const hfo_geom* g,
                       float t, int mode, float black, float white, int cz) {
    const int H = g->H, W = g->W, Si = g->in_stride, So = g->out_stride, rs = g->rs;
    const int lw = g->lw, lh = g->lh, hdr = g->hdr;
    const long N = (long)lw * lh;
    const float s12 = t, s21 = 1.0f - t;                              /* opticalFlowCalcSDR.cpp:149-150 */
    const int vo = H >> 2;                                            /* :122 */
    const int rows = H >> cz;
    const int dim_y = cz ? (H >> 1) : H;
    const float hy = cz ? 0.5f : 1.0f;
    const unsigned mid = hdr ? 32768u : 128u;
    for (int cy = 0; cy < rows; cy++) {
        for (int cx = 0; cx < W; cx++) {
            int ax = cx, ay = cy;
            const long oidx = (long)cz * H * So + (long)cy * So + cx;
            if (mode == 5 && cx < (W >> 1)) {                         /* :133-135 */
                store_el(out, hdr, oidx, load_el(A, hdr, (long)cz * H * Si + (long)cy * Si + cx));
                continue;
            } else if (mode == 6) {                                   /* :136-150 */
                const int in_rows = cy >= (vo >> cz) && cy < ((vo >> cz) + (H >> (1 + cz)));
                if (in_rows && cx < (W >> 1)) {
                    store_el(out, hdr, oidx, load_el(A, hdr, (long)cz * H * Si + (long)((cy - (vo >> cz)) << 1) * Si
                                                             + (cx << 1) + (cz ? (cx & 1) : 0)));
                    continue;
                } else if (in_rows && cx >= (W >> 1) && cx < W) {
                    ax = (cx - (W >> 1)) << 1;
                    ay = (cy - (vo >> cz)) << 1;
                } else {
                    store_el(out, hdr, oidx, cz ? mid : 0u);
                    continue;
                }
            }
            /* :153-158 */
            const int lx = cz ? ((ax >> rs) & ~1) : (ax >> rs);
            const int ly = cz ? ((ay >> rs) << 1) : (ay >> rs);
            const int ox12 = flow[(long)ly * lw + lx];
            const int oy12 = flow[N + (long)ly * lw + lx];
            const int py = clampi(ly - (oy12 >> rs), 0, lh - 1);
            const int px = clampi(lx - (ox12 >> rs), 0, lw - 1);
            const int ox21 = flow[(long)py * lw + px];
            const int oy21 = flow[N + (long)py * lw + px];
            if (mode == 4) {                                          /* :161-164 */
                const unsigned mag = (unsigned)abs(ox12) + (unsigned)abs(oy12);
                unsigned v;
                if (hdr) { v = mag << 10; if (v > 65535u) v = 65535u; }
                else     { v = mag << 2;  if (v > 255u) v = 255u; }
                store_el(out, hdr, oidx, cz ? mid : v);
                continue;
            }
            /* :167-170 */
            const int x12 = mirror_warp(ax + (int)roundf((float)ox12 * s12), W);
            const int y12 = mirror_warp(ay + (int)roundf((float)oy12 * s12 * hy), dim_y);
            const int x21 = mirror_warp(ax - (int)roundf((float)ox21 * s21), W);
            const int y21 = mirror_warp(ay - (int)roundf((float)oy21 * s21 * hy), dim_y);
            const long ia = (long)cz * H * Si + (long)y12 * Si + (cz ? (x12 & ~1) : x12) + (cz ? (cx & 1) : 0);
            const long ib = (long)cz * H * Si + (long)y21 * Si + (cz ? (x21 & ~1) : x21) + (cz ? (cx & 1) : 0);
            if (mode == 0) { store_el(out, hdr, oidx, load_el(A, hdr, ia)); continue; }   /* :172-173 */
            if (mode == 1) { store_el(out, hdr, oidx, load_el(B, hdr, ib)); continue; }   /* :174-175 */
            /* :176-183 */
            const float fa = (float)load_el(A, hdr, ia), fb = (float)load_el(B, hdr, ib);
            float bl;
            if (g_blend_flavour == 1) bl = fmaf(fa, s21, fb * s12);
            else if (g_blend_flavour == 2) bl = fmaf(fb, s12, fa * s21);
            else bl = fa * s21 + fb * s12;
            unsigned blended = (unsigned)(uint16_t)bl;
            if (mode == 3)
                blended = visualize_flow((int16_t)(-ox12), (int16_t)(-oy12), hdr ? blended : (blended & 0xFFu),
                                         cz + (cz ? (cx & 1) : 0), rs <= 2 ? 4 : 1, hdr);
            const float bk = hdr ? black * 256.0f : black, wh = hdr ? white * 256.0f : white;
            store_el(out, hdr, oidx, cz ? levels_uv((float)blended, wh, hdr) : levels_y((float)blended, bk, wh, hdr));
        }
    }
}

void hfo_warp_frames(const void* frame12, const void* frame21, const int16_t* flow, void* out,
                     const hfo_geom* g, float t, int mode, float black, float white) {
    warp_plane(frame12, frame21, flow, out, g, t, mode, black, white, 0);
    warp_plane(frame12, frame21, flow, out, g, t, mode, black, white, 1);
}

void hfo_copy_frame(const void* src, void* out, const hfo_geom* g, float black, float white) {
    const int H = g->H, W = g->W, Si = g->in_stride, So = g->out_stride, hdr = g->hdr;
    const float bk = hdr ? black * 256.0f : black, wh = hdr ? white * 256.0f : white; /* opticalFlowCalcHDR.cpp:173-174 */
    for (int cz = 0; cz < 2; cz++)
        for (int cy = 0; cy < (H >> cz); cy++)
            for (int cx = 0; cx < W; cx++) {
                const unsigned v = load_el(src, hdr, (long)cz * H * Si + (long)cy * Si + cx);
                store_el(out, hdr, (long)cz * H * So + (long)cy * So + cx,
                         cz ? levels_uv((float)v, wh, hdr) : levels_y((float)v, bk, wh, hdr));
            }
}
