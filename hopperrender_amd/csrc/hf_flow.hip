// hopperrender_amd/csrc/hf_flow.hip -- the refinement chain of calculateOpticalFlow on gfx950.
//
// Replaces (reference HopperRender/): calcDeltaSumsKernel{SDR,HDR}.h:36-191,
// determineLowestLayerKernelSDR.h:4-28, adjustOffsetArrayKernelSDR.h:4-21 and the fills of
// opticalFlowCalcSDR.cpp:68-76.  Same results, different organisation (DESIGN.md "flow chain"):
//
//  1. PHASE PLANE.  A candidate samples frame N-1 at full-resolution x = (cx << rs) + offset, i.e.
//     every 2^rs-th element starting at an arbitrary phase: 1 useful sample per 8-16 bytes.  At upload
//     each frame is re-laid out once as ONE plane of 4-byte elements (hf_kernels.h PhaseLayout)
//         PP[y][ph2][j] = top8 of  Y[y][x], Y[y][x+1], U[y>>1][x&~1], V[y>>1][x&~1],   x = mirror((j << rs) + 2*ph2)
//     with j running over [-MX, lw + MX) so the reference's edge reflection
//     (calcDeltaSumsKernelSDR.h:86-95) is baked in.  Offsets are constant inside a window (below), so
//     the samples of a run of grid pixels are CONSECUTIVE elements of one phase row: a lane fetches luma
//     and chroma of 4 pixels with ONE dword-aligned 16-byte load (v_perm_b32 drops the luma byte of the
//     other phase) and scores them with four v_sad_u8.  The grid samples of frame N are phase 0 of its own plane.
//  2. PER-WINDOW STATE.  The chain starts from zero offsets (opticalFlowCalcSDR.cpp:68-69) and every
//     update adds one value per window of a size the current size divides, so offsets, the offset bias
//     (:105-109) and the neighbour bias (:112-144) are per-window constants:
//         sum_w(cost) = (sum_w SAD) << deltaScalar + npix_w * (offsetBias + neighborBias)   (mod 2^32)
//     Offsets therefore live in one small table per level (window size) instead of per pixel.
//  3. FUSED STEPS.  For windows <= 32 one workgroup (or lane group) owns a window, and the Y step of a
//     level only needs the window's own new X offset plus LAST level's Y offsets of its neighbours, so X
//     and Y of a level run in ONE launch (5 launches for levels 32..2).  Larger windows take two
//     launches per axis (partial sums with one atomic per candidate per workgroup, then a tiny argmin).
//  4. CANDIDATE ROWS THROUGH LDS.  The 16 candidates of a Y step differ by S k + rho (S = 2^rs): within a residue class rho they read
//     the same plane rows for neighbouring grid rows, so a tile of 16 grid rows needs 95 distinct row segments (rs = 3), not 256.  Tiles
//     that lie in one window (and every 8 x 8 window by itself) copy those rows once into LDS and read the candidates from there
//     (ysads_tile_lds / ysads_win8_lds): the chain's launches are bound by what passes the L1s, and this is 20 % less of it.
#include <utility>
#include "hf_kernels.h"
#include "hf_phase_plane.h"

namespace hf {

thread_local LaunchObserver* t_launch_observer = nullptr;   // hf_kernels.h HF_LAUNCH

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// sgn(d)*d*d, d = layer - R/2 (calcDeltaSumsKernelSDR.h:70-74)
__device__ __forceinline__ int rel_offset(int layer, int R) {
    const int d = layer - (R >> 1);
    return d > 0 ? d * d : -(d * d);
}

// single reflection of calcDeltaSumsKernelSDR.h:86-95; the clamp only acts where the reference would
// index outside the frame (offsets larger than the frame) and keeps everything memory-safe
__device__ __forceinline__ int mirror_clamp(int p, int dim) {
    if (p >= dim) p = 2 * dim - p - 1; else if (p < 0) p = -p - 1;
    return clampi(p, 0, dim - 1);
}

// ------------------------------------------------------------------------------------------
// phase plane
// ------------------------------------------------------------------------------------------
// Generic build (any geometry): one workgroup per full-resolution luma row.  The luma row and its chroma row are
// read once with coalesced 16-byte loads, reduced to their top 8 bits in LDS, and written back as nph2 phase rows
// including the mirrored margins.
template <typename E>
__global__ __launch_bounds__(256) void prep_phase_kernel(const PrepBatch batch, int H, int W, int S, PhaseLayout pl) {
    const E* __restrict__ f = (const E*)batch.frame[blockIdx.y];     // blockIdx.y: frame of the batch
    uint32_t* __restrict__ pp = batch.pp[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) uint8_t row8[];   // [2][Wp]: top 8 bits of the luma row, of the chroma row
    constexpr int VEC = 16 / sizeof(E);
    const int Wp = ((W + 15) / 16) * 16;
    const int row = blockIdx.x, tid = threadIdx.x;
    for (int z = 0; z < 2; z++) {
        const E* __restrict__ src = z ? f + (size_t)H * S + (size_t)(row >> 1) * S : f + (size_t)row * S;
        uint8_t* dst8 = row8 + z * Wp;
        const bool aligned = (((uintptr_t)src) & 15) == 0;
        for (int i = tid * VEC; i < W; i += 256 * VEC) {
            if (aligned && i + VEC <= W) {
                __attribute__((aligned(16))) E v[VEC];
                *(uint4*)v = *(const uint4*)(src + i);
#pragma unroll
                for (int k = 0; k < VEC; k++) dst8[i + k] = (uint8_t)top8<E>(v[k]);
            } else {
                for (int k = 0; k < VEC && i + k < W; k++) dst8[i + k] = (uint8_t)top8<E>(src[i + k]);
            }
        }
    }
    __syncthreads();
    const int step = 1 << pl.rs;
    uint32_t* __restrict__ dst = pp + (size_t)row * pl.nph2 * pl.lwp;
    for (int t = tid; t < pl.nph2 * pl.lwp; t += 256) {
        const int ph2 = t / pl.lwp, jc = t - ph2 * pl.lwp;
        const int x = (jc - pl.mx) * step + 2 * ph2;
        const int xa = mirror_clamp(x, W), xc = xa & ~1;
        const uint32_t yb = pl.nph > 1 ? row8[mirror_clamp(x + 1, W)] : 0u;
        dst[t] = pack_element(row8[xa], yb, row8[Wp + xc], row8[Wp + xc + 1]);
    }
}

// Fast path of the plane build (no LDS): one plane_fast_task (hf_phase_plane.h) per thread.
template <typename E, int RS, bool NT>
__global__ __launch_bounds__(128) void prep_phase_fast_kernel(const PrepBatch batch, int H, int W, int S, PhaseLayout pl) {
    plane_fast_task<E, RS, 2, NT>((const E*)batch.frame[blockIdx.z], batch.pp[blockIdx.z], H, W, S, pl, (int)blockIdx.y, (int)(blockIdx.x * 128 + threadIdx.x), 0);
}

// ------------------------------------------------------------------------------------------
// per-window constants
// ------------------------------------------------------------------------------------------
// The four neighbour offsets of a window as two packed pairs of 16-bit values biased by 0x8000 (unsigned order = signed order), so that
// the neighbour term of a candidate -- four |neighbour - candidate| -- is two v_sad_u16 against the packed candidate.
struct NbPacked { uint32_t n01, n23; };
__device__ __forceinline__ NbPacked pack_nb(const int* nb) {
    NbPacked p;
    p.n01 = (((uint32_t)nb[0] + 0x8000u) & 0xFFFFu) | (((uint32_t)nb[1] + 0x8000u) << 16);
    p.n23 = (((uint32_t)nb[2] + 0x8000u) & 0xFFFFu) | (((uint32_t)nb[3] + 0x8000u) << 16);
    return p;
}
struct WinConst {
    int ox, oy;                 // offsets of this window before the level's update
    NbPacked nbx, nby;          // neighbour offsets (previous level) for the X and the Y step
    uint32_t npix;              // grid pixels of the window that lie inside the grid
};

__device__ __forceinline__ int table_at(const int16_t* __restrict__ t, const FlowLevel& L, int px, int py) {
    HF_DBG_CHECK(px >= 0 && py >= 0 && (px >> L.log2w) < L.nwx && (py >> L.log2w) < L.nwy, 100);
    return t[(py >> L.log2w) * L.nwx + (px >> L.log2w)];
}

// Level `cur`, window (wx, wy).  prev.tx == nullptr <=> first level (all offsets zero).
// use_cur_x: the X offset of this level has already been written (separate Y launch of a large window).
__device__ __forceinline__ WinConst load_win_const(const Geom& g, const FlowStep& a, int wx, int wy, bool use_cur_x) {
    WinConst w;
    const int ws = a.cur.window;
    const int x0 = wx << a.cur.log2w, y0 = wy << a.cur.log2w;
    w.ox = w.oy = 0;
    int nbx[4] = {0, 0, 0, 0}, nby[4] = {0, 0, 0, 0};
    if (a.prev.tx) {
        w.ox = table_at(a.prev.tx, a.prev, x0, y0);
        w.oy = table_at(a.prev.ty, a.prev, x0, y0);
        if (a.use_neighbors) {  // calcDeltaSumsKernelSDR.h:112-131: neighbours at +-2*window, clamped to the grid;
                                // the clamped neighbour of every pixel of a window falls into one previous-level window
            const int d = 2 * ws;
            const int xl = max(x0 - d, 0), xr = min(x0 + d, g.lw - 1);
            const int yu = max(y0 - d, 0), yd = min(y0 + d, g.lh - 1);
            nbx[0] = table_at(a.prev.tx, a.prev, x0, yd); nby[0] = table_at(a.prev.ty, a.prev, x0, yd);
            nbx[1] = table_at(a.prev.tx, a.prev, xr, y0); nby[1] = table_at(a.prev.ty, a.prev, xr, y0);
            nbx[2] = table_at(a.prev.tx, a.prev, xl, y0); nby[2] = table_at(a.prev.ty, a.prev, xl, y0);
            nbx[3] = table_at(a.prev.tx, a.prev, x0, yu); nby[3] = table_at(a.prev.ty, a.prev, x0, yu);
        }
    }
    w.nbx = pack_nb(nbx); w.nby = pack_nb(nby);
    HF_DBG_CHECK(wx >= 0 && wy >= 0 && wx < a.cur.nwx && wy < a.cur.nwy, 101);
    if (use_cur_x) w.ox = a.cur.tx[wy * a.cur.nwx + wx];
    w.npix = (uint32_t)((min(g.lw, x0 + ws) - x0) * (min(g.lh, y0 + ws) - y0));
    return w;
}

// Per-window constant part of the cost of candidate offset `cand` (short arithmetic as in the reference; |neighbour - candidate| of two
// int16 values never exceeds 16 bits, so the packed unsigned form is exact).
__device__ __forceinline__ uint32_t window_bias(int cand, bool use_nb, const NbPacked& nb, int nshift) {
    uint32_t c = (uint32_t)(cand < 0 ? -cand : cand) & 0xFFFFu;                   // offsetBias, :105-109
    if (use_nb) {
        const uint32_t cu = ((uint32_t)cand + 0x8000u) & 0xFFFFu, cc = cu | (cu << 16);
        c += __builtin_amdgcn_sad_u16(nb.n01, cc, __builtin_amdgcn_sad_u16(nb.n23, cc, 0u)) << nshift;   // :112-143
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// strip SADs: PX consecutive grid pixels of one row, all candidates of one axis
// ------------------------------------------------------------------------------------------
// PX consecutive elements of a phase row.  Element addresses are dword-aligned by construction, so this is ONE
// global_load_dwordx4 / dwordx2 on the fast path of the texture addresser (tools/ubench/gather_rate.hip: 17.5 clocks
// per wave instruction for dwordx4 at a 4-byte boundary, against 17.3 + 33.4 for the unaligned dword + dwordx2 pair
// the byte planes of round 1 needed).
template <int PX> struct __attribute__((aligned(4))) Elems { uint32_t d[PX]; };
template <int PX>
__device__ __forceinline__ Elems<PX> load_elems(const uint32_t* __restrict__ p) {
    Elems<PX> r;
    __builtin_memcpy(&r, p, sizeof(r));
    return r;
}

template <int PX>
struct Strip {
    uint32_t ref[PX];                 // frame-N samples of the strip: Y | 0 | U << 16 | V << 24 (0 outside the grid)
    uint32_t vm[PX];                  // all ones for pixels inside the grid
    int cx0, cy;
    bool any;                         // at least one pixel inside the grid
};

// FULL (here and below): the whole tile lies inside the grid and R == 16 -- no validity masks, no per-candidate tests
template <int PX, bool FULL>
__device__ __forceinline__ Strip<PX> load_strip(const Geom& g, const FlowStep& a, int cx0, int cy) {
    Strip<PX> s;
    s.cx0 = cx0; s.cy = cy;
    const int n = FULL ? PX : (cy < g.lh ? clampi(g.lw - cx0, 0, PX) : 0);
    s.any = n > 0;
#pragma unroll
    for (int i = 0; i < PX; i++) { s.ref[i] = 0u; s.vm[i] = i < n ? 0xFFFFFFFFu : 0u; }
    if (s.any) {
        const PhaseLayout& pl = a.pl;
        const int sy = cy << g.rs;   // grid samples of frame N = phase 0 of its own plane (:98-100, frame2 operands)
        HF_DBG_CHECK(cx0 >= 0 && sy >= 0 && ((size_t)sy * pl.nph2 * pl.lwp + pl.mx + cx0 + PX) * 4 <= pl.bytes, 102);
        const Elems<PX> e = load_elems<PX>(a.pp2 + (size_t)sy * pl.nph2 * pl.lwp + pl.mx + cx0);
#pragma unroll
        for (int i = 0; i < PX; i++) s.ref[i] = e.d[i] & 0xFFFF00FFu & s.vm[i];
    }
    return s;
}

// PX consecutive elements through a BUFFER load: resource (plane base + size) in SGPRs, `voff` = per-lane byte offset,
// `soff` = wave-uniform byte offset (scalar register, costs no vector instruction).
typedef unsigned flow_v2 __attribute__((ext_vector_type(2)));
typedef unsigned flow_v4 __attribute__((ext_vector_type(4)));
template <int PX>
__device__ __forceinline__ Elems<PX> buffer_elems(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, [[maybe_unused]] size_t plane_bytes) {
    HF_DBG_CHECK((size_t)voff + soff + 4 * PX <= plane_bytes && ((voff + soff) & 3u) == 0, 103);   // (the hardware would return 0 beyond the plane)
    Elems<PX> r;
    if constexpr (PX == 4) {
        const flow_v4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; r.d[3] = v.w;
    } else {
        const flow_v2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
        r.d[0] = v.x; r.d[1] = v.y;
    }
    return r;
}

// sad[cz] += sum over the strip of |dY| + |dU| + |dV| for candidate cz of `axis` (cz < R; the caller zeroes sad[] before the first strip).
// The plane offset of a candidate splits into a per-LANE part (the strip's own row / column) and a per-WINDOW part (what
// the candidate offset adds): X step  lane: row(sy + oy) * row_el + mx + cx0      window: (ph >> 1) * lwp + (c >> rs)
//                             Y step  lane: sy * row_el + (ph0 >> 1) * lwp + mx + j0   window: c * row_el
// with c = searched offset + rel(cz) (sx = cx0 << rs is a multiple of 2^rs, so (sx + c) >> rs = cx0 + (c >> rs) and the
// phase is c's alone).  UNI (every lane of the wave lies in ONE window: windows >= 16): the window part is wave-uniform
// -- it is computed on the scalar unit and rides in the load's scalar offset, so a candidate costs no address arithmetic
// on the vector ALU at all (round 1: ~13 of the ~36 VALU instructions per candidate).  Otherwise it is per lane.
// (The reference does this arithmetic in 16-bit, calcDeltaSumsKernelSDR.h:75-76; offsets are bounded by
//  iterations * 64 + 64 < 2^15, so nothing ever wraps.)
// Four candidate elements of a strip against its four frame-N samples.  PACK: the SADs of the strip's two 2-pixel halves side by side
// -- pixels 0, 1 in the low and pixels 2, 3 in the high 16 bits (each <= 2 x 765 per row) -- ADDED to t: the per-block form the SAD
// tables keep (below); otherwise the sum of all four, added to t.
template <bool PACK>
__device__ __forceinline__ uint32_t sad4(const uint32_t* c, uint32_t sel, const uint32_t* ref, uint32_t t) {
    if constexpr (PACK) {
        uint32_t hi = __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(c[2], c[2], sel), ref[2], 0u);
        hi = __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(c[3], c[3], sel), ref[3], hi);
        t = __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(c[0], c[0], sel), ref[0], t);
        t = __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(c[1], c[1], sel), ref[1], t);
        return t + (hi << 16);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) t = __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(c[i], c[i], sel), ref[i], t);
        return t;
    }
}

// R16 (here and below): the search radius is 16 although the tile may hang over the grid's edge (FULL false): the candidate loops are
// unconditional and only the validity masks remain -- the partial tiles of a launch (the bottom tile row of a 480 x 270 grid) are its
// longest waves, and with a run-time radius every candidate sits behind its own test
// NH: the candidates are fetched and scored in NH groups of 16 / NH, one after the other -- 16 / NH loads in flight per lane instead of 16:
// registers for round trips.  The table bodies (most of whose waves only reuse) take 2: their launches are as fast as their lean waves are
// many, and those are as many as the registers of the computing path allow.
template <int PX, bool UNI, bool FULL, bool PACK = false, bool R16 = FULL, int NH = 1>
__device__ __forceinline__ void strip_sads(uint32_t* sad, const Geom& g, const FlowStep& a, const Strip<PX>& s,
                                           int ox, int oy, int axis) {
    static_assert(!PACK || (PX == 4 && FULL), "the packed per-block form exists for full tiles of 4-pixel strips");
    static_assert(NH == 1 || NH == 2 || NH == 4, "groups of 16, 8 or 4 candidates");
    constexpr int NC = 16 / NH;
    const PhaseLayout& pl = a.pl;
    const int sy = s.cy << g.rs;
    const bool ragged = !FULL && (g.lw & (PX - 1)) != 0;              // kernel-uniform: some strip hangs over the right grid edge
    const int R = R16 ? 16 : a.R;
    const bool any = FULL || s.any;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.pp1, 0, (int)pl.bytes, 0x00020000);
    if (UNI) { ox = __builtin_amdgcn_readfirstlane(ox); oy = __builtin_amdgcn_readfirstlane(oy); }
    const int searched0 = axis ? oy : ox;
    const unsigned row_el = (unsigned)(pl.nph2 * pl.lwp);             // elements per full-res row
    // Y step: rows -- the reflection of calcDeltaSumsKernelSDR.h:86-95 only acts within 64 rows of the frame edge
    const int ph0 = ox & (pl.nph - 1);
    const unsigned col = (unsigned)((ph0 >> 1) * pl.lwp + pl.mx + s.cx0 + (ox >> g.rs));
    const unsigned selc = 0x03020c00u | (unsigned)(ph0 & 1);
    const int cmin = searched0 + rel_offset(0, R), cmax = searched0 + rel_offset(R - 1, R);
    const bool inside = !any || (sy + cmin >= 0 && sy + cmax <= g.H - 1);
    const bool y_direct = axis && __builtin_amdgcn_ballot_w64(!inside) == 0;
#pragma unroll
    for (int h = 0; h < NH; h++) {
        Elems<PX> c1[NC];
        uint32_t sel[NC];
        if (!axis) {
            // (the margin mx goes into the window part: it keeps that part >= 0, as a scalar buffer offset has to be)
            const unsigned lane_off = (__umul24((unsigned)mirror_clamp(sy + oy, g.H), row_el) + (unsigned)s.cx0) * 4u;
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int cz = h * NC + k;
                if (cz < R && any) {                                  // R is uniform
                    const int c = searched0 + rel_offset(cz, R);
                    const int ph = c & (pl.nph - 1);
                    const unsigned coff = (unsigned)((ph >> 1) * pl.lwp + (c >> g.rs) + pl.mx) * 4u;
                    c1[k] = UNI ? buffer_elems<PX>(rsrc, lane_off + 0u, coff, pl.bytes) : buffer_elems<PX>(rsrc, lane_off + coff, 0u, pl.bytes);
                    sel[k] = 0x03020c00u | (unsigned)(ph & 1);        // v_perm_b32: luma byte of this phase, 0, U, V
                }
            }
        } else if (y_direct) {
            const unsigned lane_off = (__umul24((unsigned)(sy + cmin), row_el) + col) * 4u;   // row of the lowest candidate
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int cz = h * NC + k;
                if (cz < R && any) {
                    const unsigned coff = __umul24((unsigned)(rel_offset(cz, R) - rel_offset(0, R)), row_el) * 4u;   // >= 0, wave-uniform
                    c1[k] = buffer_elems<PX>(rsrc, lane_off, coff, pl.bytes);
                    sel[k] = selc;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int cz = h * NC + k;
                if (cz < R && any) {
                    const int ny = mirror_clamp(sy + searched0 + rel_offset(cz, R), g.H);
                    c1[k] = buffer_elems<PX>(rsrc, (__umul24((unsigned)ny, row_el) + col) * 4u, 0u, pl.bytes);
                    sel[k] = selc;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const int cz = h * NC + k;
            uint32_t t = sad[cz];
            if constexpr (PACK) {
                t = sad4<true>(c1[k].d, sel[k], s.ref, t);
            } else if (cz < R && any) {
#pragma unroll
                for (int i = 0; i < PX; i++) {
                    uint32_t v = __builtin_amdgcn_perm(c1[k].d[i], c1[k].d[i], sel[k]);
                    if (ragged) v &= s.vm[i];
                    t = __builtin_amdgcn_sad_u8(v, s.ref[i], t);
                }
            }
            sad[cz] = t;
        }
        if constexpr (NH > 1) __builtin_amdgcn_sched_barrier(0);      // (keep the groups apart: the scheduler would hoist every load to the top)
    }
}

// ------------------------------------------------------------------------------------------
// Y step of a tile that lies in ONE window, out of LDS
// ------------------------------------------------------------------------------------------
// A Y step reads, for grid row r and candidate c, plane row ((cy0 + r) << rs) + oy + rel(c) at ONE column segment: a 16 x 16 window is
// 16 rows x 16 candidates = 256 row segments of 64 bytes, each an L2 request of its own (two 64-byte sectors at dword alignment; PMC of
// round 4: the level-16 launch pulled 1.02 x its candidate bytes through the L1s, the Y steps all of them, while the X steps of the wide
// tiles re-use their lines: 0.19 x).  But rel(c) = S k + rho (S = 2^rs) and row = S (cy0 + r + k) + oy + rho: the candidates of one residue
// class rho walk the SAME rows P_rho[j] = S (cy0 + j) + oy + rho, j = r + k -- 95 distinct rows for 16 grid rows at rs = 3, 97 at rs = 2,
// not 256.  The workgroup copies those rows once into LDS (direct-to-LDS buffer loads, 16 bytes per lane, the rows of a class
// consecutive) and every candidate is then ONE ds_read_b128 at the thread's own chunk + a compile-time row offset (threads are row-major
// over the tile: 64 lanes read 1 KB contiguous, conflict-free).  Same bytes, same order of additions as strip_sads: identical results.
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }
__host__ __device__ constexpr int rel16(int cz) { const int d = cz - 8; return d > 0 ? d * d : -(d * d); }
// RS: log2 of the frame-to-grid ratio; WROWS: grid rows of the tile; LPR: threads per tile row (4 grid pixels each); NW: waves of the workgroup
template <int RS, int WROWS, int LPR, int NW> struct YRows {
    static constexpr int S = 1 << RS;
    __host__ __device__ static constexpr int rho(int cz) { return rel16(cz) & (S - 1); }
    __host__ __device__ static constexpr int k(int cz) { return (rel16(cz) - rho(cz)) / S; }
    // classes in order of first appearance; a class is named by any of its candidates
    __host__ __device__ static constexpr bool first_of_class(int cz) { for (int i = 0; i < cz; i++) if (rho(i) == rho(cz)) return false; return true; }
    __host__ __device__ static constexpr int kmin(int cz) { int m = k(cz); for (int i = 0; i < 16; i++) if (rho(i) == rho(cz) && k(i) < m) m = k(i); return m; }
    __host__ __device__ static constexpr int kmax(int cz) { int m = k(cz); for (int i = 0; i < 16; i++) if (rho(i) == rho(cz) && k(i) > m) m = k(i); return m; }
    __host__ __device__ static constexpr int rows(int cz) { return WROWS + kmax(cz) - kmin(cz); }                       // staged rows of cz's class
    __host__ __device__ static constexpr int base(int cz) {                                                              // first staged row of cz's class
        int b = 0;
        for (int i = 0; i < 16; i++) { if (rho(i) == rho(cz)) return b; if (first_of_class(i)) b += rows(i); }
        return b;
    }
    __host__ __device__ static constexpr int total() { int b = 0; for (int i = 0; i < 16; i++) if (first_of_class(i)) b += rows(i); return b; }
    __host__ __device__ static constexpr int row0(int cz) { return base(cz) + k(cz) - kmin(cz); }                        // staged row of candidate cz for tile row 0
    __host__ __device__ static constexpr int disp(int cz) { return S * (kmin(cz) - base(cz)) + rho(cz); }                // plane row of staged row q of the class = S cy0 + oy + disp + S q
    static constexpr int kTotal = total();
    static constexpr int kCopies = (kTotal * LPR + 64 * NW - 1) / (64 * NW);                                              // copy instructions per wave (64 x 16 bytes each)
    static constexpr int kDwords = kCopies * NW * 256;                                                                    // LDS
};
// staged row q of a class holds plane row S cy0 + oy + disp + S q: for tile row r, candidate cz that must be S (cy0 + r) + oy + rel16(cz)
template <int RS> constexpr bool yrows_consistent() {
    using Y = YRows<RS, 16, 4, 1>;
    for (int cz = 0; cz < 16; cz++)
        if (Y::disp(cz) + Y::S * Y::row0(cz) != rel16(cz) || Y::row0(cz) < Y::base(cz) || Y::row0(cz) + 16 > Y::base(cz) + Y::rows(cz)) return false;
    return true;
}
static_assert(yrows_consistent<0>() && yrows_consistent<1>() && yrows_consistent<2>() && yrows_consistent<3>() && yrows_consistent<4>() &&
              YRows<3, 16, 4, 1>::kTotal == 95 && YRows<2, 16, 4, 1>::kTotal == 97, "class tables of the staged Y step");
// Dynamic LDS of a launch whose tiles may take the staged Y step (0: this rs keeps the gathers).  The launch passes exactly what its
// geometry needs: 24 KB for the partial kernel's Y launches at rs = 3, nothing for its X launches.
template <int WROWS, int LPR, int NW> constexpr size_t ystage_bytes(int rs) {
    return 4 * (size_t)(rs == 0 ? YRows<0, WROWS, LPR, NW>::kDwords : rs == 1 ? YRows<1, WROWS, LPR, NW>::kDwords : rs == 2 ? YRows<2, WROWS, LPR, NW>::kDwords :
                        rs == 3 ? YRows<3, WROWS, LPR, NW>::kDwords : rs == 4 ? YRows<4, WROWS, LPR, NW>::kDwords : 0);
}

// tid: thread of the workgroup, row-major over the tile (tile row tid / LPR); (wx0, cy0): first grid column / row of the tile; ref: the thread's
// four frame-N samples.  Workgroup-uniform call (contains a barrier when NW > 1).
template <int RS, int WROWS, int LPR, int NW>
__device__ __forceinline__ void ystage_copy(const FlowStep& a, int ox, int oy, int wx0, int cy0, int tid, uint32_t* stage) {
    using Y = YRows<RS, WROWS, LPR, NW>;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const PhaseLayout& pl = a.pl;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned row_el = (unsigned)(pl.nph2 * pl.lwp);
    const int ph0 = ox & (pl.nph - 1);
    const unsigned colbase = (unsigned)((ph0 >> 1) * pl.lwp + pl.mx + wx0 + (ox >> RS));
    const int rowbase = (cy0 << RS) + oy;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.pp1, 0, (int)pl.bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < Y::kCopies; i++) {
        const int t0 = (i * NW + wave) * 64;                    // wave-uniform
        const int t = t0 + lane;
        const int q = min(t / LPR, Y::kTotal - 1);              // (spare slots behind the last row copy it again)
        int d = 0;
        static_for<16>([&](auto CZ) {                            // class of staged row q (compile-time tables, one compare per class)
            constexpr int cz = decltype(CZ)::value;
            if constexpr (Y::first_of_class(cz)) {
                constexpr int b = Y::base(cz), dd = Y::disp(cz);
                if (cz == 0 || q >= b) d = dd;
            }
        });
        const unsigned off = (__umul24((unsigned)(rowbase + d + (q << RS)), row_el) + colbase + (unsigned)((t & (LPR - 1)) * 4)) * 4u;
        HF_DBG_CHECK(rowbase + d + (q << RS) >= 0 && (size_t)off + 16 <= pl.bytes, 108);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(stage + t0 * 4), 16, off, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): this wave's rows are in LDS
    if constexpr (NW > 1) __syncthreads();
}
// ... and the candidates of the thread whose tile-row-major index is tid (a one-wave workgroup that walks a taller tile in rounds of 64
// threads passes round * 64 + lane), out of the staged rows.
template <int RS, int WROWS, int LPR, int NW, bool PACK = false, int NH = 1>
__device__ __forceinline__ void ystage_sads(uint32_t* sad, const FlowStep& a, const uint32_t* ref, int ox, int tid, const uint32_t* stage) {
    using Y = YRows<RS, WROWS, LPR, NW>;
    const int ph0 = ox & (a.pl.nph - 1);
    const uint32_t sel = 0x03020c00u | (unsigned)(ph0 & 1);
    typedef __attribute__((address_space(3))) const flow_v4* lds_chunks;
    const lds_chunks mine = (lds_chunks)stage + tid;           // tile row r of a class lies LPR r chunks behind the class's row for tile row 0
    static_for<NH>([&](auto H) {        // (PACK: the table bodies read their candidates in NH groups, see strip_sads)
        constexpr int h = decltype(H)::value, NC = 16 / NH;
        flow_v4 c1[NC];
        static_for<NC>([&](auto K) {
            constexpr int k = decltype(K)::value, cz = h * NC + k, chunk = Y::row0(cz) * LPR;
            c1[k] = mine[chunk];
        });
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const uint32_t c[4] = {c1[k][0], c1[k][1], c1[k][2], c1[k][3]};
            sad[h * NC + k] = sad4<PACK>(c, sel, ref, sad[h * NC + k]);
        }
        if constexpr (NH > 1) __builtin_amdgcn_sched_barrier(0);
    });
}
template <int RS, int WROWS, int LPR, int NW, bool PACK = false, int NH = 1>
__device__ __forceinline__ void ysads_tile_lds(uint32_t* sad, const FlowStep& a, const uint32_t* ref, int ox, int oy, int wx0, int cy0, int tid, uint32_t* stage) {
    ystage_copy<RS, WROWS, LPR, NW>(a, ox, oy, wx0, cy0, tid, stage);
    ystage_sads<RS, WROWS, LPR, NW, PACK, NH>(sad, a, ref, ox, tid, stage);
}

// The staged form applies to full tiles (R = 16) of planes with rs <= 4 whose candidate rows need no reflection.  ox, oy: the window's.
template <int WROWS, int LPR, int NW, bool PACK = false, int NH = 1>
__device__ __forceinline__ bool ysads_tile_try(uint32_t* sad, const Geom& g, const FlowStep& a, const uint32_t* ref, int ox, int oy, int wx0, int cy0, int tid, uint32_t* stage) {
    ox = __builtin_amdgcn_readfirstlane(ox); oy = __builtin_amdgcn_readfirstlane(oy);
    wx0 = __builtin_amdgcn_readfirstlane(wx0); cy0 = __builtin_amdgcn_readfirstlane(cy0);
    if (g.rs < 0 || g.rs > 4 || !a.y_rows_lds) return false;   // (kernel-uniform; such a launch has no dynamic LDS)
    if ((cy0 << g.rs) + oy + rel16(0) < 0 || ((cy0 + WROWS - 1) << g.rs) + oy + rel16(15) > g.H - 1) return false;
    switch (g.rs) {
        case 0: ysads_tile_lds<0, WROWS, LPR, NW, PACK, NH>(sad, a, ref, ox, oy, wx0, cy0, tid, stage); break;
        case 1: ysads_tile_lds<1, WROWS, LPR, NW, PACK, NH>(sad, a, ref, ox, oy, wx0, cy0, tid, stage); break;
        case 2: ysads_tile_lds<2, WROWS, LPR, NW, PACK, NH>(sad, a, ref, ox, oy, wx0, cy0, tid, stage); break;
        case 3: ysads_tile_lds<3, WROWS, LPR, NW, PACK, NH>(sad, a, ref, ox, oy, wx0, cy0, tid, stage); break;
        default: ysads_tile_lds<4, WROWS, LPR, NW, PACK, NH>(sad, a, ref, ox, oy, wx0, cy0, tid, stage); break;
    }
    return true;
}

// (the test of ysads_tile_try alone, and the run-time rs as a compile-time constant for code that stages in its own order)
template <int WROWS>
__device__ __forceinline__ bool ytile_can_stage(const Geom& g, const FlowStep& a, int oy, int cy0) {
    if (g.rs < 0 || g.rs > 4 || !a.y_rows_lds) return false;
    return !((cy0 << g.rs) + oy + rel16(0) < 0 || ((cy0 + WROWS - 1) << g.rs) + oy + rel16(15) > g.H - 1);
}
template <class F>
__device__ __forceinline__ void with_rs(int rs, F&& f) {
    switch (rs) {
        case 0: f(std::integral_constant<int, 0>{}); break;
        case 1: f(std::integral_constant<int, 1>{}); break;
        case 2: f(std::integral_constant<int, 2>{}); break;
        case 3: f(std::integral_constant<int, 3>{}); break;
        default: f(std::integral_constant<int, 4>{}); break;
    }
}

// The same for 8 x 8 windows, four per wave (Map<8>: 16 lanes = one window, lane i of it = tile row i / 2, column group i & 1): every window
// has its own offsets, so each 16-lane group copies ITS 63 (rs = 3) / 73 (rs = 2) rows of 32 bytes -- 128 row segments per window before --
// with per-lane addresses.  A copy instruction lands lane-contiguous in LDS: slot t of window w lies at 1,024 (t / 16) + 256 w + 16 (t % 16).
template <int RS, bool PACK = false, int NH = 1>
__device__ __forceinline__ void ysads_win8_lds(uint32_t* sad, const FlowStep& a, const uint32_t* ref, int ox, int oy, int cx0, int cy, int lane, uint32_t* stage) {
    using Y = YRows<RS, 8, 2, 1>;
    constexpr int kCopies = (Y::kTotal * 2 + 15) / 16;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const PhaseLayout& pl = a.pl;
    const int i16 = lane & 15;
    const unsigned row_el = (unsigned)(pl.nph2 * pl.lwp);
    const int ph0 = ox & (pl.nph - 1);
    const unsigned colbase = (unsigned)((ph0 >> 1) * pl.lwp + pl.mx + (cx0 - (i16 & 1) * 4) + (ox >> RS));
    const int rowbase = ((cy - (i16 >> 1)) << RS) + oy;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.pp1, 0, (int)pl.bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < kCopies; i++) {
        const int t = i * 16 + i16;
        const int q = min(t >> 1, Y::kTotal - 1);
        int d = 0;
        static_for<16>([&](auto CZ) {
            constexpr int cz = decltype(CZ)::value;
            if constexpr (Y::first_of_class(cz)) {
                constexpr int b = Y::base(cz), dd = Y::disp(cz);
                if (cz == 0 || q >= b) d = dd;
            }
        });
        const unsigned off = (__umul24((unsigned)(rowbase + d + (q << RS)), row_el) + colbase + (unsigned)((t & 1) * 4)) * 4u;
        HF_DBG_CHECK(rowbase + d + (q << RS) >= 0 && (size_t)off + 16 <= pl.bytes, 109);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(stage + i * 256), 16, off, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0); one wave = one workgroup
    const uint32_t sel = 0x03020c00u | (unsigned)(ph0 & 1);
    typedef __attribute__((address_space(3))) const flow_v4* lds_chunks;
    const lds_chunks mine = (lds_chunks)stage + (lane >> 4) * 16;   // the window's 16 chunks of copy 0
    static_for<NH>([&](auto H) {
        constexpr int h = decltype(H)::value, NC = 16 / NH;
        flow_v4 c1[NC];
        static_for<NC>([&](auto K) {
            constexpr int k = decltype(K)::value, cz = h * NC + k, c2 = Y::row0(cz) * 2, A = c2 & 15;
            const int u = A + i16;                              // slot c2 + i16 = copy (c2 / 16) + (u / 16), chunk u % 16
            c1[k] = mine[(c2 >> 4) * 64 + u + (A != 0 && u >= 16 ? 48 : 0)];
        });
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const uint32_t c[4] = {c1[k][0], c1[k][1], c1[k][2], c1[k][3]};
            sad[h * NC + k] = sad4<PACK>(c, sel, ref, sad[h * NC + k]);
        }
        if constexpr (NH > 1) __builtin_amdgcn_sched_barrier(0);
    });
}
template <int RS> constexpr size_t win8_stage_bytes_of() { return (size_t)((YRows<RS, 8, 2, 1>::kTotal * 2 + 15) / 16) * 1024; }
constexpr size_t win8_stage_bytes(int rs) {
    return rs == 0 ? win8_stage_bytes_of<0>() : rs == 1 ? win8_stage_bytes_of<1>() : rs == 2 ? win8_stage_bytes_of<2>() : rs == 3 ? win8_stage_bytes_of<3>() :
           rs == 4 ? win8_stage_bytes_of<4>() : 0;
}

// cx0, cy, ox, oy: the lane's own (per window).  Taken only if all four windows of the wave need no reflection.
template <bool PACK = false, int NH = 1>
__device__ __forceinline__ bool ysads_win8_try(uint32_t* sad, const Geom& g, const FlowStep& a, const uint32_t* ref, int ox, int oy, int cx0, int cy, int lane, uint32_t* stage) {
    if (g.rs < 0 || g.rs > 4 || !a.y_rows_lds) return false;
    const int cy0 = cy - ((lane & 15) >> 1);
    const bool outside = (cy0 << g.rs) + oy + rel16(0) < 0 || ((cy0 + 7) << g.rs) + oy + rel16(15) > g.H - 1;
    if (__builtin_amdgcn_ballot_w64(outside) != 0) return false;
    switch (g.rs) {
        case 0: ysads_win8_lds<0, PACK, NH>(sad, a, ref, ox, oy, cx0, cy, lane, stage); break;
        case 1: ysads_win8_lds<1, PACK, NH>(sad, a, ref, ox, oy, cx0, cy, lane, stage); break;
        case 2: ysads_win8_lds<2, PACK, NH>(sad, a, ref, ox, oy, cx0, cy, lane, stage); break;
        case 3: ysads_win8_lds<3, PACK, NH>(sad, a, ref, ox, oy, cx0, cy, lane, stage); break;
        default: ysads_win8_lds<4, PACK, NH>(sad, a, ref, ox, oy, cx0, cy, lane, stage); break;
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// cross-lane reduction
// ------------------------------------------------------------------------------------------
// Value of lane (l ^ M) -- on the vector ALU's own cross-lane paths, never through the LDS crossbar (__shfl_xor = ds_bpermute_b32: an LDS
// instruction and ~100 cycles of dependent latency per exchange; a level launch made ~50 of them in a row, a good part of its skeleton):
// M = 1, 2 DPP quad_perm; M = 4 two bank-masked DPP row shifts (lanes with bit 2 clear read lane + 4, the others lane - 4); M = 8 DPP
// row_ror:8; M = 16 / 32 gfx950's v_permlane16_swap / v_permlane32_swap of the value with itself + one select.
template <int CTRL, int BANKS>
__device__ __forceinline__ uint32_t dpp_into(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, BANKS, false);
}
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "xor mask inside a wave");
    if constexpr (M == 1) return dpp_into<0xB1, 0xF>(v, v);                     // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_into<0x4E, 0xF>(v, v);                // quad_perm [2,3,0,1]
    else if constexpr (M == 4) return dpp_into<0x114, 0xA>(dpp_into<0x104, 0x5>(v, v), v);   // row_shl:4 into banks 0, 2; row_shr:4 into banks 1, 3
    else if constexpr (M == 8) return dpp_into<0x128, 0xF>(v, v);               // row_ror:8
    else if constexpr (M == 16) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? r[0] : r[1]; }
    else { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? r[0] : r[1]; }
}

// One butterfly level: lanes l and l^M exchange halves of their NV values; afterwards each lane
// holds NV/2 values, each the pair-sum of one candidate.  Lanes with bit M set keep the upper half.
// M = 16 / 32: the swap instructions ARE this exchange -- v_permlane32_swap(x, y) leaves x = [x.lo, y.lo], y = [x.hi, y.hi], so x + y is
// x summed over the pair in the lower lanes and y summed over the pair in the upper ones: one swap + one add per value, no selects.
template <int NV, int M>
__device__ __forceinline__ void butterfly_level(uint32_t* v, int lane) {
#pragma unroll
    for (int k = 0; k < NV / 2; k++) {
        if constexpr (M == 32) {
            const auto r = __builtin_amdgcn_permlane32_swap(v[k], v[k + NV / 2], false, false);
            v[k] = r[0] + r[1];
        } else if constexpr (M == 16) {
            const auto r = __builtin_amdgcn_permlane16_swap(v[k], v[k + NV / 2], false, false);
            v[k] = r[0] + r[1];
        } else {
            const bool hi = (lane & M) != 0;
            const uint32_t send = hi ? v[k] : v[k + NV / 2];
            const uint32_t keep = hi ? v[k + NV / 2] : v[k];
            v[k] = keep + lane_xor<M>(send, lane);
        }
    }
}

struct Best { uint32_t sum; int cz; };
__device__ __forceinline__ void best_min(Best& b, uint32_t s, int cz) {   // first minimum wins (strict '<', determineLowestLayerKernelSDR.h:19-24)
    if (s < b.sum || (s == b.sum && cz < b.cz)) { b.sum = s; b.cz = cz; }
}
template <int M>
__device__ __forceinline__ void best_xor(Best& b, int lane) {
    const uint32_t s = lane_xor<M>(b.sum, lane);
    const int c = (int)lane_xor<M>((uint32_t)b.cz, lane);
    best_min(b, s, c);
}
template <int M>
__device__ __forceinline__ uint32_t or_xor(uint32_t v, int lane) { return v | lane_xor<M>(v, lane); }

// Reduces sad[16] over an aligned group of G lanes (G = 2, 4, 16, 64).  Afterwards each lane owns NOWN
// consecutive candidates starting at `first` with their group totals in sad[0..NOWN).
template <int G> struct Owned;
template <> struct Owned<64> { static constexpr int n = 1; };
template <> struct Owned<16> { static constexpr int n = 1; };
template <> struct Owned<4> { static constexpr int n = 4; };
template <> struct Owned<2> { static constexpr int n = 8; };
template <> struct Owned<1> { static constexpr int n = 16; };

// XM (G == 2 only): the partner of a lane is lane ^ XM
template <int G, int XM = 1>
__device__ __forceinline__ int group_reduce(uint32_t* sad, int lane) {
    if constexpr (G == 1) {
        return 0;
    } else if constexpr (G == 2) {
        butterfly_level<16, XM>(sad, lane);
        return (lane & XM) ? 8 : 0;
    } else if constexpr (G == 64) {
        butterfly_level<16, 32>(sad, lane); butterfly_level<8, 16>(sad, lane);
        butterfly_level<4, 8>(sad, lane);   butterfly_level<2, 4>(sad, lane);
        sad[0] += lane_xor<2>(sad[0], lane);
        sad[0] += lane_xor<1>(sad[0], lane);
        return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    } else if constexpr (G == 16) {
        butterfly_level<16, 8>(sad, lane); butterfly_level<8, 4>(sad, lane);
        butterfly_level<4, 2>(sad, lane);  butterfly_level<2, 1>(sad, lane);
        return ((lane >> 3) & 1) * 8 + ((lane >> 2) & 1) * 4 + ((lane >> 1) & 1) * 2 + (lane & 1);
    } else {
        static_assert(G == 4, "lane groups of 1, 2, 4, 16 or 64");
        butterfly_level<16, 2>(sad, lane); butterfly_level<8, 1>(sad, lane);
        return ((lane >> 1) & 1) * 8 + (lane & 1) * 4;
    }
}

// argmin over all candidates of the group; every lane of the group returns the same winner.
// `captured` (optional) receives the full cost sum of candidate cap_cz.
template <int G, bool R16, int XM = 1>
__device__ __forceinline__ int group_argmin(const uint32_t* tot, int first, const FlowStep& a, int searched0,
                                            const NbPacked& nb, uint32_t npix, int cap_cz, bool want_cap, uint32_t& captured, int lane) {
    const int R = R16 ? 16 : a.R;
    Best b{0xFFFFFFFFu, 16};
    uint32_t cap = 0;
#pragma unroll
    for (int k = 0; k < Owned<G>::n; k++) {
        const int cz = first + k;
        if (cz < R) {
            const int cand = (int)(int16_t)(searched0 + rel_offset(cz, R));
            const uint32_t sum = (tot[k] << a.delta_scalar) + npix * window_bias(cand, a.use_neighbors, nb, a.neighbor_scalar);
            if (k == 0 || sum < b.sum) { b.sum = sum; b.cz = cz; }   // ascending cz inside the lane: strict '<' keeps the first minimum (:19-24)
            if (cz == cap_cz) cap = sum;
        }
    }
    if (G == 64) { best_xor<4>(b, lane); best_xor<8>(b, lane); best_xor<16>(b, lane); best_xor<32>(b, lane); }
    else if (G == 16) { best_xor<1>(b, lane); best_xor<2>(b, lane); best_xor<4>(b, lane); best_xor<8>(b, lane); }
    else if (G == 4) { best_xor<1>(b, lane); best_xor<2>(b, lane); }
    else if (G == 2) { best_xor<XM>(b, lane); }
    if (want_cap) {
        if (G == 64) { cap = or_xor<4>(cap, lane); cap = or_xor<8>(cap, lane); cap = or_xor<16>(cap, lane); cap = or_xor<32>(cap, lane); }
        else if (G == 16) { cap = or_xor<1>(cap, lane); cap = or_xor<2>(cap, lane); cap = or_xor<4>(cap, lane); cap = or_xor<8>(cap, lane); }
        else if (G == 4) { cap = or_xor<1>(cap, lane); cap = or_xor<2>(cap, lane); }
        else if (G == 2) { cap = or_xor<XM>(cap, lane); }
        captured = cap;
    }
    return b.cz;
}

// Resolves a.pend for the pending-level window that contains grid pixel (px, py): returns the window's
// offset on the pending axis AFTER that step.  Must be called by whole waves whose lanes all lie in the
// same pending window (true for every 32x32 tile); each aligned 16-lane group computes it redundantly.
// `origin_leader`: this thread stores the result (exactly one thread per pending window does).
__device__ __forceinline__ int resolve_pending(const Geom& g, const FlowStep& a, int px, int py, int lane, bool origin_leader) {
    const PendingArgmin& p = a.pend;
    const int wx = px >> p.lvl.log2w, wy = py >> p.lvl.log2w, w = wy * p.lvl.nwx + wx;
    const int x0 = wx << p.lvl.log2w, y0 = wy << p.lvl.log2w;
    int searched0 = 0;
    if (p.lvl_prev.tx) searched0 = table_at(p.axis ? p.lvl_prev.ty : p.lvl_prev.tx, p.lvl_prev, x0, y0);
    const uint32_t npix = (uint32_t)((min(g.lw, x0 + p.lvl.window) - x0) * (min(g.lh, y0 + p.lvl.window) - y0));
    const int cz = lane & 15;
    Best b{0xFFFFFFFFu, 16};
    uint32_t mine = 0;
    HF_DBG_CHECK(wx >= 0 && wy >= 0 && wx < p.lvl.nwx && wy < p.lvl.nwy, 104);
    if (cz < a.R) {
        const int cand = (int)(int16_t)(searched0 + rel_offset(cz, a.R));
        mine = (p.sums[w * 16 + cz] << a.delta_scalar) + npix * ((uint32_t)(cand < 0 ? -cand : cand) & 0xFFFFu);  // no neighbour term (level < 4)
        b.sum = mine; b.cz = cz;
    }
    best_xor<1>(b, lane); best_xor<2>(b, lane); best_xor<4>(b, lane); best_xor<8>(b, lane);
    const int value = (int)(int16_t)(searched0 + rel_offset(b.cz, a.R));
    if (p.capture_delta && w == 0) {   // opticalFlowCalcSDR.cpp:91-94
        uint32_t cap = cz == (a.R >> 1) - 1 ? mine : 0u;
        cap = or_xor<1>(cap, lane); cap = or_xor<2>(cap, lane); cap = or_xor<4>(cap, lane); cap = or_xor<8>(cap, lane);
        if (origin_leader) *a.total_delta = cap / a.delta_divisor;
    }
    if (origin_leader) (p.axis ? p.lvl.ty : p.lvl.tx)[w] = (int16_t)value;
    return value;
}

// ------------------------------------------------------------------------------------------
// workgroup -> tile mapping, XCD-aware
// ------------------------------------------------------------------------------------------
// The chain kernels use 1-D grids.  Consecutive workgroup ids go to consecutive XCDs (MI355X_MICROARCH.md "Workgroup
// dispatch": id % 8), each with its own L2, and neighbouring tiles read overlapping candidate rows of the phase planes
// (a Y step reaches +-8 grid rows).  With the natural order every XCD ends up fetching nearly the whole footprint of a
// step (measured with rocprofv3 FETCH_SIZE: 6.7 MB for a launch whose footprint is ~2 MB, 129 MB per 2160p chain).
// So unit u = (id % 8) * per_xcd + id / 8: XCD k works on the k-th contiguous eighth of the units, ordered
// (pair, tile row, tile column, wave) -- a band of one pair's grid, or whole pairs of a batch.
// Kernel-argument form of a FlowBatch.  A launch carries at most 4 KB of arguments and a FlowStep is 248 bytes, but the
// members of a batch differ only in their 13 buffer pointers: the launch gets ONE FlowStep (member 0's) plus the pointers
// of every member (104 bytes each: 32 members = 3.3 KB), and a workgroup rebuilds its member's FlowStep in scalar registers.
struct FlowPtrs {           // what differs between the members of a batch: six allocations (every table / sum pointer is member 0's + a rebase)
    const uint32_t *pp1, *pp2;
    int16_t* tables;
    uint32_t *sums, *total_delta, *sadtab, *still_count;
};
struct FlowBatchArgs {
    int n;
    FastDiv tiles, tiles_x;              // unit index -> (pair, tile row, tile column): scalar divisions (hf_kernels.h)
    FlowStep common;
    FlowPtrs m[kMaxFlowBatch];
};
static_assert(sizeof(FlowBatchArgs) + sizeof(Geom) <= 4096, "kernel arguments of a batched chain launch");
// waves_per_tile: what decode_tile divides the unit index by first (SPLIT launches); the dividers see at most tiles x members (+ the
// grid's padding to a multiple of 8)
static FlowBatchArgs pack_batch(const FlowBatch& b, int tiles_x, int tiles_y) {
    FlowBatchArgs k;
    k.n = b.n;
    const uint64_t max_v = (uint64_t)tiles_x * tiles_y * b.n + 8;
    k.tiles = make_fastdiv((uint32_t)(tiles_x * tiles_y), max_v); k.tiles_x = make_fastdiv((uint32_t)tiles_x, (uint64_t)tiles_x * tiles_y);
    k.common = b.s[0];
    for (int i = 0; i < b.n; i++) {
        const FlowStep& f = b.s[i];
        k.m[i] = FlowPtrs{f.pp1, f.pp2, f.tables_base, f.sums_base, f.total_delta, f.sadtab, f.still_count};
    }
    return k;
}
__device__ __forceinline__ FlowStep member_step(const FlowBatchArgs& k, int i) {
    FlowStep a = k.common;
    const FlowPtrs& p = k.m[i];
    a.pp1 = p.pp1; a.pp2 = p.pp2; a.total_delta = p.total_delta; a.sadtab = p.sadtab; a.still_count = p.still_count;
    const ptrdiff_t dt = p.tables - a.tables_base, ds = p.sums - a.sums_base;   // (scalar arithmetic; null stays null)
    auto rt = [dt](int16_t* x) { return x ? x + dt : x; };
    a.cur.tx = rt(a.cur.tx); a.cur.ty = rt(a.cur.ty); a.prev.tx = rt(a.prev.tx); a.prev.ty = rt(a.prev.ty);
    a.prev2.tx = rt(a.prev2.tx); a.prev2.ty = rt(a.prev2.ty);
    a.pend.lvl.tx = rt(a.pend.lvl.tx); a.pend.lvl.ty = rt(a.pend.lvl.ty); a.pend.lvl_prev.tx = rt(a.pend.lvl_prev.tx); a.pend.lvl_prev.ty = rt(a.pend.lvl_prev.ty);
    a.sums = a.sums ? a.sums + ds : a.sums;
    a.pend.sums = a.pend.sums ? a.pend.sums + ds : a.pend.sums;
    a.tables_base = p.tables; a.sums_base = p.sums;
    return a;
}

struct TileId { int pair, tx, ty, wave; bool valid; };
template <int WPT>   // waves per tile: 1, 2 or 4
__device__ __forceinline__ TileId decode_tile(const FlowBatchArgs& k, int tiles_x, int tiles_y) {
    const int n_tiles = tiles_x * tiles_y;
    const int total = n_tiles * WPT * k.n;
    const int per = (total + 7) >> 3;                       // the grid is exactly 8 * per workgroups
    const int u = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    TileId t;
    t.valid = u < total;
    t.wave = u & (WPT - 1);
    const int v = u / WPT;
    t.pair = min((int)fastdiv((uint32_t)v, k.tiles), k.n - 1);
    const int tile = v - t.pair * n_tiles;                  // (only meaningful for valid units)
    t.ty = (int)fastdiv((uint32_t)max(tile, 0), k.tiles_x);
    t.tx = tile - t.ty * tiles_x;
    return t;
}
inline int xcd_grid(int tiles_x, int tiles_y, int waves_per_tile, int n_pairs) {
    return ((tiles_x * tiles_y * waves_per_tile * n_pairs + 7) >> 3) << 3;
}

// ------------------------------------------------------------------------------------------
// lane -> strip mapping: every window of size WS is an aligned, contiguous lane group
// ------------------------------------------------------------------------------------------
// PX consecutive grid pixels x NR consecutive grid rows per lane; G lanes per window (G == 2: the partner is lane ^ XM).
template <int WS> struct Map;
template <> struct Map<32> {   // workgroup tile 32x32 = one window; wave = 8 rows
    static constexpr int PX = 4, NR = 1, G = 64, XM = 1, TW = 32, TH = 32, WAVES = 4, RM = 8;   // RM: lane ^ RM = the same columns one grid row up / down
    __device__ static void at(int tid, int& x, int& y) { x = (tid & 7) * 4; y = tid >> 3; }
};
template <> struct Map<16> {   // wave = one 16x16 window; workgroup = 2x2 windows
    static constexpr int PX = 4, NR = 1, G = 64, XM = 1, TW = 32, TH = 32, WAVES = 4, RM = 4;
    __device__ static void at(int tid, int& x, int& y) {
        const int w = tid >> 6, l = tid & 63;
        x = (w & 1) * 16 + (l & 3) * 4; y = (w >> 1) * 16 + (l >> 2);
    }
};
template <> struct Map<8> {    // 16 lanes = one 8x8 window; wave = 2x2 windows
    static constexpr int PX = 4, NR = 1, G = 16, XM = 1, TW = 32, TH = 32, WAVES = 4, RM = 2;
    __device__ static void at(int tid, int& x, int& y) {
        const int w = tid >> 6, l = tid & 63, gi = l >> 4, i = l & 15;
        x = (w & 1) * 16 + (gi & 1) * 8 + (i & 1) * 4; y = (w >> 1) * 16 + (gi >> 1) * 8 + (i >> 1);
    }
};
// The two finest levels: a lane owns a 4 x 2 / 2 x 2 BLOCK -- both rows' SADs add up in the lane's own registers.  (Round 3: one row per
// lane, so a 2 x 2 window was two lanes with two pixels each -- twice the waves, each paying the full per-wave skeleton of window
// constants, bias terms and argmin for half the pixels, plus a butterfly; the level-2 and level-4 launches were the two most expensive
// of the chain.)
template <> struct Map<4> {    // 2 lanes (l, l ^ 8) = one 4x4 window; wave = 8x4 windows = 32 px x 16 rows; a 32x32 tile is TWO waves
    static constexpr int PX = 4, NR = 2, G = 2, XM = 8, TW = 32, TH = 32, WAVES = 2;
    __device__ static void at(int tid, int& x, int& y) {
        const int w = tid >> 6, l = tid & 63;
        x = (l & 7) * 4; y = w * 16 + (l >> 4) * 4 + ((l >> 3) & 1) * 2;
    }
};
template <> struct Map<2> {    // one lane = one 2x2 window; wave = 16x4 windows = 32 px x 8 rows; workgroup tile 32x32
    static constexpr int PX = 2, NR = 2, G = 1, XM = 1, TW = 32, TH = 32, WAVES = 4;
    __device__ static void at(int tid, int& x, int& y) {
        const int w = tid >> 6, l = tid & 63;
        x = (l & 15) * 2; y = w * 8 + (l >> 4) * 2;
    }
};

// The same two levels with ONE ROW per lane (round 3's mapping): twice the waves for the same pixels.  A batch of many pairs pays for the
// extra per-wave skeleton; a single pair (the reference's own operating mode: one live stream) leaves most of the device idle, and there
// twice the waves finish sooner -- 8.7 / 7.8 us instead of 12.4 / 10.2 us for the level-2 / level-4 launch of one 480 x 270 pair.
template <int WS> struct MapRow : Map<WS> {};
template <> struct MapRow<4> {   // 4 lanes = one 4x4 window; wave = 4x4 windows
    static constexpr int PX = 4, NR = 1, G = 4, XM = 1, TW = 32, TH = 32, WAVES = 4, RM = 1;
    __device__ static void at(int tid, int& x, int& y) {
        const int w = tid >> 6, l = tid & 63, gi = l >> 2;
        x = (w & 1) * 16 + (gi & 3) * 4; y = (w >> 1) * 16 + (gi >> 2) * 4 + (l & 3);
    }
};
template <> struct MapRow<2> {   // 2 lanes (l, l ^ 1) = one 2x2 window; wave = 8x4 windows; workgroup tile 16x32
    static constexpr int PX = 2, NR = 1, G = 2, XM = 1, TW = 16, TH = 32, WAVES = 4;
    __device__ static void at(int tid, int& x, int& y) {
        const int w = tid >> 6, l = tid & 63, gi = l >> 1;
        x = (gi & 7) * 2; y = w * 8 + (gi >> 3) * 2 + (l & 1);
    }
};
template <int WS, bool ROWS1> struct MapSel { using type = Map<WS>; };
template <int WS> struct MapSel<WS, true> { using type = MapRow<WS>; };

// One level, X step then Y step, windows <= 32.
// SPLIT (windows <= 16, where a window never spans waves): the four waves of a tile are four one-wave workgroups
// (TileId::wave).  A 480x270 grid has only 135 tiles for 256 CUs; split, every CU's L1 takes a share of the
// candidate rows' cache lines (a Y step pulls ~16 x 8 row segments per wave, 32 useful bytes per 128-byte line).
template <int WS, bool SPLIT, bool FULL, bool ROWS1, bool R16 = FULL, int NH = 1>
__device__ __forceinline__ void flow_level_small_body(const Geom& g, const FlowStep& a, const TileId& tile, uint32_t (*s_part)[4][16], [[maybe_unused]] uint32_t* s_rows) {
    using M = typename MapSel<WS, ROWS1>::type;
    constexpr int PX = M::PX, G = M::G;
    const int R = R16 ? 16 : a.R;
    const int tid = SPLIT ? (int)(tile.wave * 64 + threadIdx.x) : (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int lx, ly;
    M::at(tid, lx, ly);
    const int cx0 = tile.tx * M::TW + lx, cy = tile.ty * M::TH + ly;
    if constexpr (!FULL && SPLIT) {      // a wave tile that lies outside the grid altogether (the lower half of a bottom tile): nothing to do, no window starts in it
        if (__builtin_amdgcn_ballot_w64(cx0 < g.lw && cy < g.lh) == 0) return;
    }
    const int wx = cx0 >> a.cur.log2w, wy = cy >> a.cur.log2w;
    const bool win_in = FULL || ((wx << a.cur.log2w) < g.lw && (wy << a.cur.log2w) < g.lh);   // whole lane group agrees

    WinConst wc{};
    if (win_in) wc = load_win_const(g, a, wx, wy, false);
    if (a.pend.active) {   // the last large-window step (Y of the previous level) is resolved here
        const int tx0 = tile.tx * M::TW, ty0 = tile.ty * M::TH;   // tile origin: inside the grid
        const bool leader = tid == 0 && (tx0 & (a.pend.lvl.window - 1)) == 0 && (ty0 & (a.pend.lvl.window - 1)) == 0;
        const int v = resolve_pending(g, a, tx0, ty0, lane, leader);
        if (a.pend.axis) wc.oy = v; else wc.ox = v;
    }
    Strip<PX> strip[M::NR];
#pragma unroll
    for (int r = 0; r < M::NR; r++) strip[r] = load_strip<PX, FULL>(g, a, cx0, cy + r);
    const int cap_cz = (R >> 1) - 1;
    uint32_t captured = 0;
    int off[2] = {wc.ox, wc.oy};

#pragma unroll
    for (int axis = 0; axis < 2; axis++) {
        uint32_t sad[16];
#pragma unroll
        for (int cz = 0; cz < 16; cz++) sad[cz] = 0u;
        bool from_lds = false;
        if constexpr (WS == 16 && FULL && SPLIT) {          // one wave = one window
            if (axis == 1) from_lds = ysads_tile_try<16, 4, 1>(sad, g, a, strip[0].ref, off[0], off[1], cx0, cy, lane, s_rows);   // (lane 0 holds the window's origin)
        } else if constexpr (WS == 8 && FULL && SPLIT) {    // one wave = four windows
            if (axis == 1) from_lds = ysads_win8_try(sad, g, a, strip[0].ref, off[0], off[1], cx0, cy, lane, s_rows);
        } else if constexpr (WS == 32 && FULL) {           // the workgroup = one window
            if (axis == 1) from_lds = ysads_tile_try<32, 8, 4>(sad, g, a, strip[0].ref, off[0], off[1], cx0 - lx, cy - ly, tid, s_rows);
        }
        if (!from_lds) {
#pragma unroll
            for (int r = 0; r < M::NR; r++) strip_sads<PX, G == 64, FULL, false, R16, NH>(sad, g, a, strip[r], off[0], off[1], axis);
        }
        int first = group_reduce<G, M::XM>(sad, lane);
        if constexpr (WS == 32) {   // four waves share the window
            if ((lane & 3) == 0) s_part[axis][wave][first] = sad[0];
            __syncthreads();
            sad[0] = s_part[axis][0][first] + s_part[axis][1][first] + s_part[axis][2][first] + s_part[axis][3][first];
        }
        const int best = group_argmin<G, R16, M::XM>(sad, first, a, off[axis], axis ? wc.nby : wc.nbx, wc.npix, cap_cz,
                                                      axis == 0 && a.capture_delta, captured, lane);
        off[axis] = (int)(int16_t)(off[axis] + rel_offset(best, R));   // adjustOffsetArrayKernelSDR.h:13-19
    }

    const bool leader = WS == 32 ? tid == 0 : G == 2 ? (lane & M::XM) == 0 : (lane & (G - 1)) == 0;
    HF_DBG_CHECK(!win_in || (wx >= 0 && wy >= 0 && wx < a.cur.nwx && wy < a.cur.nwy), 105);
    if (win_in && leader) {
        a.cur.tx[wy * a.cur.nwx + wx] = (int16_t)off[0];
        a.cur.ty[wy * a.cur.nwx + wx] = (int16_t)off[1];
        if (a.capture_delta && wx == 0 && wy == 0) *a.total_delta = captured / a.delta_divisor;   // opticalFlowCalcSDR.cpp:91-94
        if (WS == 32 && a.still_count && off[0] == wc.ox && off[1] == wc.oy) atomicAdd(a.still_count, 1u);   // content hint (hf_calc.hip)
    }
}

// ------------------------------------------------------------------------------------------
// SAD TABLES: exact reuse of candidate SADs across steps
// ------------------------------------------------------------------------------------------
// The reference recomputes every candidate SAD at every one of its 16 steps (opticalFlowCalcSDR.cpp:72-111, calcDeltaSumsKernelSDR.h:61-101).
// But the per-pixel SAD of candidate cz depends only on the pixel, the axis and the window's (ox, oy) BEFORE the step -- the bias terms are
// per-window constants (restructuring 2 above) -- so a window of step s samples exactly the positions step s - 2 sampled (same axis, parent
// level) whenever the two steps in between, s - 2 and s - 1, both chose d = 0 for it:
//     X step of level k:  X and Y of level k - 1 chose 0 for the parent window                (zx && zy)
//     Y step of level k:  Y of level k - 1 chose 0 for the parent, X of level k for the window (zy && own X choice == 0)
// (what a window chose is read off the level tables: parent's offset == grandparent's).  Once the global motion is found that is most
// windows (bench scene: 74-96 % of the pixels at every step of levels 16 .. 2, tests/flow_reuse_model.py).  So every step that COMPUTES
// candidate SADs leaves them per 2 x 2 grid block -- the finest window -- in a table per axis: 16 candidates x u16 (a block's SAD is at
// most 4 x 765) = 32 bytes per block, 1.04 MB per pair and axis at 480 x 270; a window that may reuse sums its blocks' vectors -- 8 bytes per
// pixel of coalesced reads instead of 64 bytes per pixel of gathers -- and leaves the entries as they are: they stay valid until somebody's
// offsets change, and whoever's do recomputes and refreshes them.  Lane l of a window holds the vector of ONE block either way (computed:
// its strip's two 2-pixel halves + the row partner's, transposed; reused: loaded), and ONE packed butterfly sums them: u16 pairs stay exact
// up to 16 blocks (48,960), wider sums continue in 32 bits -- the same integers as the per-pixel order of the reference (sums of
// non-negative terms, no wrap before the shift by deltaScalar; wrap-around of (sum << delta) + npix * bias as before).
// Applies to tiles that lie fully inside the grid at the full search radius (the only ones whose SADs have a regular block form); the
// first small level of a chain always computes.  HF_FLAG_NO_SAD_REUSE: no tables, every step computes (A-B, tests).

// v[8]: candidates (2j, 2j + 1) of one block per lane as u16 pairs.  Sums them over the G lanes of a window; afterwards each lane owns
// Owned<G>::n consecutive candidates starting at the returned index, totals in tot[].
template <int G, int XM>
__device__ __forceinline__ int packed_reduce(uint32_t* v, uint32_t* tot, int lane) {
    if constexpr (G == 64) {
        butterfly_level<8, 32>(v, lane); butterfly_level<4, 16>(v, lane); butterfly_level<2, 8>(v, lane);   // 8 blocks per half
        const uint32_t w = v[0] + lane_xor<4>(v[0], lane);                                               // 16 blocks: <= 48,960
        uint32_t t = (lane & 4) ? w >> 16 : w & 0xFFFFu;
        t += lane_xor<2>(t, lane);
        t += lane_xor<1>(t, lane);
        tot[0] = t;
        return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    } else if constexpr (G == 16) {
        butterfly_level<8, 8>(v, lane); butterfly_level<4, 4>(v, lane); butterfly_level<2, 2>(v, lane);
        const uint32_t w = v[0] + lane_xor<1>(v[0], lane);
        tot[0] = (lane & 1) ? w >> 16 : w & 0xFFFFu;
        return ((lane >> 3) & 1) * 8 + ((lane >> 2) & 1) * 4 + ((lane >> 1) & 1) * 2 + (lane & 1);
    } else if constexpr (G == 4) {
        butterfly_level<8, 2>(v, lane); butterfly_level<4, 1>(v, lane);
        tot[0] = v[0] & 0xFFFFu; tot[1] = v[0] >> 16; tot[2] = v[1] & 0xFFFFu; tot[3] = v[1] >> 16;
        return ((lane >> 1) & 1) * 8 + (lane & 1) * 4;
    } else {
        static_assert(G == 2, "windows of 64, 16, 4 or 2 lanes");
        butterfly_level<8, XM>(v, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) { tot[2 * j] = v[j] & 0xFFFFu; tot[2 * j + 1] = v[j] >> 16; }
        return (lane & XM) ? 8 : 0;
    }
}

__device__ __forceinline__ void unpack_u16x8(const flow_v4& q, uint32_t* o) {
    o[0] = q.x & 0xFFFFu; o[1] = q.x >> 16; o[2] = q.y & 0xFFFFu; o[3] = q.y >> 16;
    o[4] = q.z & 0xFFFFu; o[5] = q.z >> 16; o[6] = q.w & 0xFFFFu; o[7] = q.w >> 16;
}

// candidate groups of the table kernels (strip_sads NH): 8 candidates in flight.  Groups of 4: 3 % slower stand-alone, no faster in the pipeline.
constexpr int kTabGroups = 2;
// One level, X step then Y step, windows <= 32, of windows in tiles that lie fully inside the grid at R == 16, with the SAD tables.
// A reusing window is a short chain of dependent memory rounds -- kernel arguments, then ONE round with everything whose address does not
// depend on data: the window's constants, the grandparent's offsets and the table vectors of BOTH axes (whether they may be used is only
// known later) -- and ~40 registers; a computing window keeps 16 candidates x 16 bytes in flight (100-170 registers) and several rounds.
// (cx0, cy): first grid pixel of the lane; (lx, ly): its position inside the tile (WS == 32 needs the tile origin).
template <int WS, bool SPLIT, bool ROWS1>
__device__ __forceinline__ void flow_level_tab_body(const Geom& g, const FlowStep& a, int cx0, int cy, int lx, int ly, int tid, const TileId& tile,
                                                    uint32_t (*s_part)[4][16], [[maybe_unused]] uint32_t* s_rows) {
    using M = typename MapSel<WS, ROWS1>::type;
    constexpr int PX = M::PX, G = M::G, NR = M::NR, NOWN = Owned<G>::n;
    constexpr int NV = (PX == 4 && NR == 2) ? 4 : (PX == 2 && NR == 1) ? 1 : 2;     // 16-byte vectors of a lane per axis
    const int wave = tid >> 6, lane = tid & 63;
    const int wx = cx0 >> a.cur.log2w, wy = cy >> a.cur.log2w;

    WinConst wc = load_win_const(g, a, wx, wy, false);
    // the lane's block(s) of the tables: 4 x 1 strips pair up with the row partner (even rows keep the left block, odd rows the right one);
    // the two lanes of a MapRow<2> window take half a block each
    const int nblk = a.sad_nbx * a.sad_nby;
    const int blk = (cy >> 1) * a.sad_nbx + (cx0 >> 1) + ((PX == 4 && NR == 1) ? (cy & 1) : 0);
    HF_DBG_CHECK(blk >= 0 && blk + (PX == 4 && NR == 2 ? 1 : 0) < nblk, 110);
    flow_v4* const tab0 = (flow_v4*)a.sadtab + (size_t)blk * 2 + ((PX == 2 && NR == 1 && (lane & M::XM)) ? 1 : 0);
    const size_t tab_axis = (size_t)nblk * 2;
    // what the parent window chose at the previous level: its offsets against its own parent's
    bool zx = false, zy = false;
    flow_v4 pre[2][NV];
    if (WS < 32 && a.sad_read) {
        int gx = 0, gy = 0;
        if (a.prev2.tx) {
            gx = table_at(a.prev2.tx, a.prev2, wx << a.cur.log2w, wy << a.cur.log2w);
            gy = table_at(a.prev2.ty, a.prev2, wx << a.cur.log2w, wy << a.cur.log2w);
        }
#pragma unroll
        for (int ax = 0; ax < 2; ax++)
#pragma unroll
            for (int v = 0; v < NV; v++) pre[ax][v] = tab0[ax * tab_axis + v];
        zx = wc.ox == gx; zy = wc.oy == gy;
    }
    if (a.pend.active) {   // the last large-window step (Y of the previous level) is resolved here
        const int tx0 = tile.tx * M::TW, ty0 = tile.ty * M::TH;
        const bool leader = tid == 0 && (tx0 & (a.pend.lvl.window - 1)) == 0 && (ty0 & (a.pend.lvl.window - 1)) == 0;
        const int v = resolve_pending(g, a, tx0, ty0, lane, leader);
        if (a.pend.axis) wc.oy = v; else wc.ox = v;
    }
    const int cap_cz = 7;
    uint32_t captured = 0;
    int off[2] = {wc.ox, wc.oy};

#pragma unroll
    for (int axis = 0; axis < 2; axis++) {
        bool reuse = zy && (axis == 0 ? zx : off[0] == wc.ox);
        if constexpr (G == 64) reuse = __builtin_amdgcn_readfirstlane((int)reuse) != 0;      // one window per wave
        if (a.counters) {     // diagnostics (hf_debug_counters): windows of this wave, those that reuse
            const bool lead = WS == 32 ? tid == 0 : G == 2 ? (lane & M::XM) == 0 : (lane & (G - 1)) == 0;
            const uint64_t mw = __builtin_amdgcn_ballot_w64(lead), mr = __builtin_amdgcn_ballot_w64(lead && reuse);
            if (lane == 0 && mw) {
                uint32_t* const c = a.counters + kCounterLevels + 4 * (a.level_index & 15) + 2 * axis;
                atomicAdd(c, (uint32_t)__builtin_popcountll(mw));
                if (mr) atomicAdd(c + 1, (uint32_t)__builtin_popcountll(mr));
            }
        }
        flow_v4* const tab = tab0 + axis * tab_axis;
        uint32_t tot[NOWN];
        int first = 0;
        if constexpr (PX == 4 && NR == 1) {
            uint32_t W[8];
            if (!reuse) {
                Strip<PX> strip[NR];          // frame-N samples of the lane: only windows that compute read them
#pragma unroll
                for (int r = 0; r < NR; r++) strip[r] = load_strip<PX, true>(g, a, cx0, cy + r);
                uint32_t p[16];
#pragma unroll
                for (int cz = 0; cz < 16; cz++) p[cz] = 0u;
                bool from_lds = false;
                if constexpr (WS == 16 && SPLIT) {
                    if (axis == 1) from_lds = ysads_tile_try<16, 4, 1, true, kTabGroups>(p, g, a, strip[0].ref, off[0], off[1], cx0, cy, lane, s_rows);
                } else if constexpr (WS == 8 && SPLIT) {
                    if (axis == 1) from_lds = ysads_win8_try<true, kTabGroups>(p, g, a, strip[0].ref, off[0], off[1], cx0, cy, lane, s_rows);
                } else if constexpr (WS == 32) {
                    if (axis == 1) from_lds = ysads_tile_try<32, 8, 4, true, kTabGroups>(p, g, a, strip[0].ref, off[0], off[1], cx0 - lx, cy - ly, tid, s_rows);
                }
                if (!from_lds) strip_sads<4, G == 64, true, true, true, kTabGroups>(p, g, a, strip[0], off[0], off[1], axis);
                const uint32_t selw = (cy & 1) ? 0x07060302u : 0x05040100u;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t q0 = p[2 * j] + lane_xor<M::RM>(p[2 * j], lane);               // both rows of the two blocks: left | right << 16
                    const uint32_t q1 = p[2 * j + 1] + lane_xor<M::RM>(p[2 * j + 1], lane);
                    W[j] = __builtin_amdgcn_perm(q1, q0, selw);                                  // this lane's block: candidates 2j, 2j + 1
                }
                if (a.sad_write) { tab[0] = flow_v4{W[0], W[1], W[2], W[3]}; tab[1] = flow_v4{W[4], W[5], W[6], W[7]}; }
            } else {
                const flow_v4 lo = pre[axis][0], hi = pre[axis][1];
                W[0] = lo.x; W[1] = lo.y; W[2] = lo.z; W[3] = lo.w; W[4] = hi.x; W[5] = hi.y; W[6] = hi.z; W[7] = hi.w;
            }
            first = packed_reduce<G, M::XM>(W, tot, lane);
        } else if constexpr (PX == 4) {      // Map<4>: a 4 x 2 block pair per lane
            uint32_t V[8];
            if (!reuse) {
                Strip<PX> strip[NR];          // frame-N samples of the lane: only windows that compute read them
#pragma unroll
                for (int r = 0; r < NR; r++) strip[r] = load_strip<PX, true>(g, a, cx0, cy + r);
                uint32_t p[16];
#pragma unroll
                for (int cz = 0; cz < 16; cz++) p[cz] = 0u;
#pragma unroll
                for (int r = 0; r < NR; r++) strip_sads<4, false, true, true, true, kTabGroups>(p, g, a, strip[r], off[0], off[1], axis);
                uint32_t W0[8], W1[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    W0[j] = __builtin_amdgcn_perm(p[2 * j + 1], p[2 * j], 0x05040100u);
                    W1[j] = __builtin_amdgcn_perm(p[2 * j + 1], p[2 * j], 0x07060302u);
                    V[j] = W0[j] + W1[j];
                }
                if (a.sad_write) {
                    tab[0] = flow_v4{W0[0], W0[1], W0[2], W0[3]}; tab[1] = flow_v4{W0[4], W0[5], W0[6], W0[7]};
                    tab[2] = flow_v4{W1[0], W1[1], W1[2], W1[3]}; tab[3] = flow_v4{W1[4], W1[5], W1[6], W1[7]};
                }
            } else {
                const flow_v4 a0 = pre[axis][0], a1 = pre[axis][1], b0 = pre[axis][2 % NV], b1 = pre[axis][3 % NV];
                V[0] = a0.x + b0.x; V[1] = a0.y + b0.y; V[2] = a0.z + b0.z; V[3] = a0.w + b0.w;
                V[4] = a1.x + b1.x; V[5] = a1.y + b1.y; V[6] = a1.z + b1.z; V[7] = a1.w + b1.w;
            }
            first = packed_reduce<G, M::XM>(V, tot, lane);
        } else if constexpr (NR == 2) {      // Map<2>: the lane IS a block
            if (!reuse) {
                Strip<PX> strip[NR];          // frame-N samples of the lane: only windows that compute read them
#pragma unroll
                for (int r = 0; r < NR; r++) strip[r] = load_strip<PX, true>(g, a, cx0, cy + r);
#pragma unroll
                for (int cz = 0; cz < 16; cz++) tot[cz] = 0u;
#pragma unroll
                for (int r = 0; r < NR; r++) strip_sads<PX, false, true, false, true, kTabGroups>(tot, g, a, strip[r], off[0], off[1], axis);
            } else {
                unpack_u16x8(pre[axis][0], tot); unpack_u16x8(pre[axis][1 % NV], tot + 8);
            }
        } else {                             // MapRow<2>: two lanes, a row each
            if (!reuse) {
                Strip<PX> strip[NR];          // frame-N samples of the lane: only windows that compute read them
#pragma unroll
                for (int r = 0; r < NR; r++) strip[r] = load_strip<PX, true>(g, a, cx0, cy + r);
                uint32_t sad[16];
#pragma unroll
                for (int cz = 0; cz < 16; cz++) sad[cz] = 0u;
                strip_sads<PX, false, true, false, true, kTabGroups>(sad, g, a, strip[0], off[0], off[1], axis);
                butterfly_level<16, M::XM>(sad, lane);
#pragma unroll
                for (int k = 0; k < 8; k++) tot[k] = sad[k];
            } else {
                unpack_u16x8(pre[axis][0], tot);
            }
            first = (lane & M::XM) ? 8 : 0;
        }
        if constexpr (WS == 32) {   // four waves share the window
            if ((lane & 3) == 0) s_part[axis][wave][first] = tot[0];
            __syncthreads();
            tot[0] = s_part[axis][0][first] + s_part[axis][1][first] + s_part[axis][2][first] + s_part[axis][3][first];
        }
        const int best = group_argmin<G, true, M::XM>(tot, first, a, off[axis], axis ? wc.nby : wc.nbx, wc.npix, cap_cz,
                                                      axis == 0 && a.capture_delta, captured, lane);
        off[axis] = (int)(int16_t)(off[axis] + rel_offset(best, 16));   // adjustOffsetArrayKernelSDR.h:13-19
    }

    const bool leader = WS == 32 ? tid == 0 : G == 2 ? (lane & M::XM) == 0 : (lane & (G - 1)) == 0;
    HF_DBG_CHECK(wx >= 0 && wy >= 0 && wx < a.cur.nwx && wy < a.cur.nwy, 105);
    if (leader) {
        a.cur.tx[wy * a.cur.nwx + wx] = (int16_t)off[0];
        a.cur.ty[wy * a.cur.nwx + wx] = (int16_t)off[1];
        if (WS == 32 && a.still_count && off[0] == wc.ox && off[1] == wc.oy) atomicAdd(a.still_count, 1u);   // content hint (hf_calc.hip)
        if (a.capture_delta && wx == 0 && wy == 0) *a.total_delta = captured / a.delta_divisor;   // opticalFlowCalcSDR.cpp:91-94
    }
}

// ------------------------------------------------------------------------------------------
// Level 32 of a batch: ONE WAVE per 32 x 32 window
// ------------------------------------------------------------------------------------------
// The four-wave workgroup of flow_level_small_kernel<32> needs four free wave slots, its registers on all four SIMDs and its candidate
// rows' LDS on one CU at the same moment.  Alone on the device that costs nothing; beside the other queues of a throughput pipeline, whose
// kernels take every slot the moment it frees, it was the chain's most stretched launch by far (timelines of round 6: 20-25 us alone,
// 65-185 us inside the pipelines of the 1080p workloads; the one-wave launches of levels 16 .. 2: 2 x).  Here a wave walks its window in
// four rounds of 64 threads x 4 pixels (8 rows each): the sums of the rounds add up in the lanes' registers where the four waves exchanged
// theirs through LDS behind a barrier -- the same integer additions in another order -- and the Y step's candidate rows are staged once per
// window by the one wave.  A quarter of the waves, each four times as long: for batches (launch_flow_level_small), where there are enough.
template <bool FULL, bool R16, int NH>
__device__ __forceinline__ void flow_level32_wave_body(const Geom& g, const FlowStep& a, const TileId& tile, [[maybe_unused]] uint32_t* s_rows) {
    using M = Map<32>;
    const int R = R16 ? 16 : a.R;
    const int lane = (int)threadIdx.x;
    int lx, ly;
    M::at(lane, lx, ly);                                       // round v: the same columns, rows ly + 8 v
    const int tx0 = tile.tx * M::TW, ty0 = tile.ty * M::TH;    // = the window's origin (inside the grid)
    const int cx0 = tx0 + lx;
    const int wx = tx0 >> a.cur.log2w, wy = ty0 >> a.cur.log2w;
    WinConst wc = load_win_const(g, a, wx, wy, false);
    if (a.pend.active) {   // the last large-window step (Y of the previous level) is resolved here
        const bool leader = lane == 0 && (tx0 & (a.pend.lvl.window - 1)) == 0 && (ty0 & (a.pend.lvl.window - 1)) == 0;
        const int v = resolve_pending(g, a, tx0, ty0, lane, leader);
        if (a.pend.axis) wc.oy = v; else wc.ox = v;
    }
    const int cap_cz = (R >> 1) - 1;
    uint32_t captured = 0;
    int off[2] = {wc.ox, wc.oy};
#pragma unroll
    for (int axis = 0; axis < 2; axis++) {
        uint32_t sad[16];
#pragma unroll
        for (int cz = 0; cz < 16; cz++) sad[cz] = 0u;
        bool staged = false;
        if constexpr (FULL) if (axis == 1) {
            staged = ytile_can_stage<32>(g, a, __builtin_amdgcn_readfirstlane(off[1]), ty0);
            if (staged) with_rs(g.rs, [&](auto RS) { ystage_copy<decltype(RS)::value, 32, 8, 1>(a, __builtin_amdgcn_readfirstlane(off[0]), __builtin_amdgcn_readfirstlane(off[1]), tx0, ty0, lane, s_rows); });
        }
#pragma unroll 1
        for (int v = 0; v < 4; v++) {
            const Strip<4> strip = load_strip<4, FULL>(g, a, cx0, ty0 + ly + 8 * v);
            if (staged) with_rs(g.rs, [&](auto RS) { ystage_sads<decltype(RS)::value, 32, 8, 1, false, NH>(sad, a, strip.ref, __builtin_amdgcn_readfirstlane(off[0]), v * 64 + lane, s_rows); });
            else strip_sads<4, true, FULL, false, R16, NH>(sad, g, a, strip, off[0], off[1], axis);
        }
        const int first = group_reduce<64, 1>(sad, lane);
        const int best = group_argmin<64, R16, 1>(sad, first, a, off[axis], axis ? wc.nby : wc.nbx, wc.npix, cap_cz,
                                                  axis == 0 && a.capture_delta, captured, lane);
        off[axis] = (int)(int16_t)(off[axis] + rel_offset(best, R));   // adjustOffsetArrayKernelSDR.h:13-19
    }
    HF_DBG_CHECK(wx >= 0 && wy >= 0 && wx < a.cur.nwx && wy < a.cur.nwy, 105);
    if (lane == 0) {
        a.cur.tx[wy * a.cur.nwx + wx] = (int16_t)off[0];
        a.cur.ty[wy * a.cur.nwx + wx] = (int16_t)off[1];
        if (a.capture_delta && wx == 0 && wy == 0) *a.total_delta = captured / a.delta_divisor;   // opticalFlowCalcSDR.cpp:91-94
        if (a.still_count && off[0] == wc.ox && off[1] == wc.oy) atomicAdd(a.still_count, 1u);   // content hint (hf_calc.hip)
    }
}

// ... with the SAD tables (full tiles at R == 16): level 32 is the chain's first table level -- it always computes, and leaves both tables.
__device__ __forceinline__ void flow_level32_wave_tab_body(const Geom& g, const FlowStep& a, const TileId& tile, [[maybe_unused]] uint32_t* s_rows) {
    using M = Map<32>;
    const int lane = (int)threadIdx.x;
    int lx, ly;
    M::at(lane, lx, ly);
    const int tx0 = tile.tx * M::TW, ty0 = tile.ty * M::TH;
    const int cx0 = tx0 + lx, cy_0 = ty0 + ly;
    const int wx = tx0 >> a.cur.log2w, wy = ty0 >> a.cur.log2w;
    WinConst wc = load_win_const(g, a, wx, wy, false);
    // the lane's block of round 0 (even rows keep the left block of the strip, odd rows the right one); round v lies 4 v block rows below
    const int nblk = a.sad_nbx * a.sad_nby;
    const int blk = (cy_0 >> 1) * a.sad_nbx + (cx0 >> 1) + (cy_0 & 1);
    HF_DBG_CHECK(blk >= 0 && blk + 12 * a.sad_nbx < nblk, 110);
    flow_v4* const tab0 = (flow_v4*)a.sadtab + (size_t)blk * 2;
    const size_t tab_axis = (size_t)nblk * 2, tab_round = (size_t)a.sad_nbx * 8;
    if (a.pend.active) {   // the last large-window step (Y of the previous level) is resolved here
        const bool leader = lane == 0 && (tx0 & (a.pend.lvl.window - 1)) == 0 && (ty0 & (a.pend.lvl.window - 1)) == 0;
        const int v = resolve_pending(g, a, tx0, ty0, lane, leader);
        if (a.pend.axis) wc.oy = v; else wc.ox = v;
    }
    const int cap_cz = 7;
    uint32_t captured = 0;
    int off[2] = {wc.ox, wc.oy};
    const uint32_t selw = (cy_0 & 1) ? 0x07060302u : 0x05040100u;
#pragma unroll
    for (int axis = 0; axis < 2; axis++) {
        if (a.counters && lane == 0) atomicAdd(a.counters + kCounterLevels + 4 * (a.level_index & 15) + 2 * axis, 1u);   // diagnostics: one window, never reusing
        flow_v4* const tab = tab0 + axis * tab_axis;
        bool staged = false;
        if (axis == 1) {
            staged = ytile_can_stage<32>(g, a, __builtin_amdgcn_readfirstlane(off[1]), ty0);
            if (staged) with_rs(g.rs, [&](auto RS) { ystage_copy<decltype(RS)::value, 32, 8, 1>(a, __builtin_amdgcn_readfirstlane(off[0]), __builtin_amdgcn_readfirstlane(off[1]), tx0, ty0, lane, s_rows); });
        }
        uint32_t total = 0;
        int first = 0;
#pragma unroll 1
        for (int v = 0; v < 4; v++) {
            const Strip<4> strip = load_strip<4, true>(g, a, cx0, cy_0 + 8 * v);
            uint32_t p[16];
#pragma unroll
            for (int cz = 0; cz < 16; cz++) p[cz] = 0u;
            if (staged) with_rs(g.rs, [&](auto RS) { ystage_sads<decltype(RS)::value, 32, 8, 1, true, kTabGroups>(p, a, strip.ref, __builtin_amdgcn_readfirstlane(off[0]), v * 64 + lane, s_rows); });
            else strip_sads<4, true, true, true, true, kTabGroups>(p, g, a, strip, off[0], off[1], axis);
            uint32_t W[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t q0 = p[2 * j] + lane_xor<M::RM>(p[2 * j], lane);               // both rows of the two blocks: left | right << 16
                const uint32_t q1 = p[2 * j + 1] + lane_xor<M::RM>(p[2 * j + 1], lane);
                W[j] = __builtin_amdgcn_perm(q1, q0, selw);                                  // this lane's block: candidates 2j, 2j + 1
            }
            if (a.sad_write) { tab[v * tab_round] = flow_v4{W[0], W[1], W[2], W[3]}; tab[v * tab_round + 1] = flow_v4{W[4], W[5], W[6], W[7]}; }
            uint32_t t1[1];
            first = packed_reduce<64, 1>(W, t1, lane);
            total += t1[0];
        }
        const uint32_t tot[1] = {total};
        const int best = group_argmin<64, true, 1>(tot, first, a, off[axis], axis ? wc.nby : wc.nbx, wc.npix, cap_cz,
                                                   axis == 0 && a.capture_delta, captured, lane);
        off[axis] = (int)(int16_t)(off[axis] + rel_offset(best, 16));   // adjustOffsetArrayKernelSDR.h:13-19
    }
    HF_DBG_CHECK(wx >= 0 && wy >= 0 && wx < a.cur.nwx && wy < a.cur.nwy, 105);
    if (lane == 0) {
        a.cur.tx[wy * a.cur.nwx + wx] = (int16_t)off[0];
        a.cur.ty[wy * a.cur.nwx + wx] = (int16_t)off[1];
        if (a.still_count && off[0] == wc.ox && off[1] == wc.oy) atomicAdd(a.still_count, 1u);   // content hint (hf_calc.hip)
        if (a.capture_delta && wx == 0 && wy == 0) *a.total_delta = captured / a.delta_divisor;   // opticalFlowCalcSDR.cpp:91-94
    }
}

template <bool TABK>
__global__ __launch_bounds__(64) void flow_level32_wave_kernel(const Geom g, const FlowBatchArgs batch) {
    const TileId tile = decode_tile<1>(batch, (g.lw + 31) / 32, (g.lh + 31) / 32);
    if (!tile.valid) return;
    const FlowStep a = member_step(batch, tile.pair);
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rows[];   // candidate rows of the Y step (launch_flow_level_small)
    const bool full = a.R == 16 && (tile.tx + 1) * 32 <= g.lw && (tile.ty + 1) * 32 <= g.lh;
    if constexpr (TABK) {      // (a.sadtab, a.R == 16 and a.sad_write)
        if (full) flow_level32_wave_tab_body(g, a, tile, s_rows);
        else flow_level32_wave_body<false, true, kTabGroups>(g, a, tile, s_rows);
    }
    else if (full) flow_level32_wave_body<true, true, 1>(g, a, tile, s_rows);
    else if (a.R == 16) flow_level32_wave_body<false, true, 1>(g, a, tile, s_rows);
    else flow_level32_wave_body<false, false, 1>(g, a, tile, s_rows);
}

// TABK: the launch's chain keeps SAD tables at R == 16 -- full tiles take the table body, tiles over the grid's edge the generic one, both with
// the candidates in groups of 8 (kTabGroups).  The kernel's register allocation is that of its fattest body, so the launches without tables
// (16 candidates in flight everywhere) are instantiations of their own.
template <int WS, bool SPLIT, bool ROWS1 = false, bool TABK = false>
__global__ __launch_bounds__(SPLIT ? 64 : 256) void flow_level_small_kernel(const Geom g, const FlowBatchArgs batch) {
    using M = typename MapSel<WS, ROWS1>::type;
    const TileId tile = decode_tile<SPLIT ? M::WAVES : 1>(batch, (g.lw + M::TW - 1) / M::TW, (g.lh + M::TH - 1) / M::TH);
    if (!tile.valid) return;
    const FlowStep a = member_step(batch, tile.pair);
    static_assert(!SPLIT || WS <= 16, "a 32x32 window is shared by the four waves of a workgroup");
    __shared__ uint32_t s_part[SPLIT ? 1 : 2][4][16];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rows[];   // the launch's dynamic LDS: candidate rows of the Y step (launch_flow_level_small)
    // Workgroup-uniform choice: tiles that lie inside the grid with the full search radius (all but the last tile row /
    // column once the governor has settled at 16) run a body without validity masks and per-candidate tests.
    // (Registers: 16 candidates x 16 bytes in flight per row = ~100 per lane for the one-row levels, 140-170 for the block levels.  Capping
    //  the one-row levels and the partial kernel at 72 -- 8 candidates in flight -- lets a chain wave fit beside the period warp's five waves
    //  per SIMD: inside the pipeline the warp launch then got 11 % shorter and the chain 39 % longer, the same frames/s; DESIGN.md appendix D.)
    const bool full = a.R == 16 && (tile.tx + 1) * M::TW <= g.lw && (tile.ty + 1) * M::TH <= g.lh;
    // SAD tables: tiles whose 32 x 32 tile lies inside the grid (the 16-wide tiles of MapRow<2> answer for the tile around them: its
    // entries are what the previous level left)
    if constexpr (TABK) {      // (launch_flow_level_small: a.sadtab, a.R == 16 and a.sad_read || a.sad_write)
        const bool tab = full && (M::TW == 32 || ((tile.tx * M::TW) | 31) < g.lw);
        if (tab) {
            const int tid = SPLIT ? (int)(tile.wave * 64 + threadIdx.x) : (int)threadIdx.x;
            int lx, ly;
            M::at(tid, lx, ly);
            flow_level_tab_body<WS, SPLIT, ROWS1>(g, a, tile.tx * M::TW + lx, tile.ty * M::TH + ly, lx, ly, tid, tile, s_part, s_rows);
        }
        else if (full) flow_level_small_body<WS, SPLIT, true, ROWS1, true, kTabGroups>(g, a, tile, s_part, s_rows);
        else flow_level_small_body<WS, SPLIT, false, ROWS1, true, kTabGroups>(g, a, tile, s_part, s_rows);
    }
    else if (full) flow_level_small_body<WS, SPLIT, true, ROWS1>(g, a, tile, s_part, s_rows);
    else if (a.R == 16) flow_level_small_body<WS, SPLIT, false, ROWS1, true>(g, a, tile, s_part, s_rows);
    else flow_level_small_body<WS, SPLIT, false, ROWS1>(g, a, tile, s_part, s_rows);
}

// Windows > 32, one axis: raw SAD sums of a tile -> one atomic per candidate.
// WPB = waves per workgroup.  Fewer waves per workgroup = more workgroups for the 256 CUs (a 480x270 grid has 135
// 32x32 tiles) at the price of more atomics on the same R addresses of each window.
template <int WPB, bool FULL, bool R16 = FULL>
__device__ __forceinline__ void flow_big_partial_body(const Geom& g, const FlowStep& a, const TileId& tile, uint32_t (*s_part)[16], [[maybe_unused]] uint32_t* s_rows) {
    const int R = R16 ? 16 : a.R;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // wave = 64 grid pixels x 4 rows: the 16 lanes the texture addresser handles together read 256 contiguous bytes of ONE
    // phase row (a 16-lane group that spans several rows costs one L1 tag lookup per row and 64-byte block)
    const int lx = (tid & 15) * 4, ly = tid >> 4;
    const int tx0 = tile.tx * 64, ty0 = tile.ty * (4 * WPB);   // 64 x 4 WPB divides every window > 32: the tile lies in one window
    const int cx0 = tx0 + lx, cy = ty0 + ly;
    const int wx = tx0 >> a.cur.log2w, wy = ty0 >> a.cur.log2w;
    int ox = 0, oy = 0;
    if (a.prev.tx) {
        ox = table_at(a.prev.tx, a.prev, wx << a.cur.log2w, wy << a.cur.log2w);
        oy = table_at(a.prev.ty, a.prev, wx << a.cur.log2w, wy << a.cur.log2w);
    }
    if (a.pend.active) {   // argmin of the previous large-window step, taken here instead of in a launch of its own
        const bool leader = tid == 0 && (tx0 & (a.pend.lvl.window - 1)) == 0 && (ty0 & (a.pend.lvl.window - 1)) == 0;
        const int v = resolve_pending(g, a, tx0, ty0, lane, leader);
        if (a.pend.axis) oy = v; else ox = v;
    } else if (a.axis == 1) {
        ox = a.cur.tx[wy * a.cur.nwx + wx];
    }
    const Strip<4> strip = load_strip<4, FULL>(g, a, cx0, cy);
    uint32_t sad[16];
#pragma unroll
    for (int cz = 0; cz < 16; cz++) sad[cz] = 0u;
    bool from_lds = false;
    if constexpr (FULL && WPB == 4) if (a.axis == 1) from_lds = ysads_tile_try<16, 16, 4>(sad, g, a, strip.ref, ox, oy, tx0, ty0, tid, s_rows);
    if (!from_lds) strip_sads<4, true, FULL, false, R16>(sad, g, a, strip, ox, oy, a.axis);
    const int first = group_reduce<64>(sad, lane);
    HF_DBG_CHECK(wx >= 0 && wy >= 0 && wx < a.cur.nwx && wy < a.cur.nwy, 106);
    uint32_t* dst = &a.sums[(wy * a.cur.nwx + wx) * 16];
    if constexpr (WPB == 1) {
        if ((lane & 3) == 0 && first < R) atomicAdd(dst + first, sad[0]);
    } else {
        if ((lane & 3) == 0) s_part[wave][first] = sad[0];
        __syncthreads();
        if (tid < R) {
            uint32_t t = 0;
#pragma unroll
            for (int w = 0; w < WPB; w++) t += s_part[w][tid];
            atomicAdd(dst + tid, t);
        }
    }
}

template <int WPB>
__global__ __launch_bounds__(64 * WPB) void flow_big_partial_kernel(const Geom g, const FlowBatchArgs batch) {
    constexpr int TW = 64, TH = 4 * WPB;
    const TileId tile = decode_tile<1>(batch, (g.lw + TW - 1) / TW, (g.lh + TH - 1) / TH);
    if (!tile.valid) return;
    const FlowStep a = member_step(batch, tile.pair);
    __shared__ uint32_t s_part[WPB][16];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rows[];   // Y launches: candidate rows (launch_flow_big_partial)
    const bool full = a.R == 16 && (tile.tx + 1) * TW <= g.lw && (tile.ty + 1) * TH <= g.lh;   // see flow_level_small_kernel
    if (full) flow_big_partial_body<WPB, true>(g, a, tile, s_part, s_rows);
    else if (a.R == 16) flow_big_partial_body<WPB, false, true>(g, a, tile, s_part, s_rows);
    else flow_big_partial_body<WPB, false>(g, a, tile, s_part, s_rows);
}

// Windows > 32, one axis: 16 lanes per window finish the sums, pick the winner, update the table.
__global__ __launch_bounds__(256) void flow_big_argmin_kernel(const Geom g, const FlowBatchArgs batch) {
    const FlowStep a = member_step(batch, (int)blockIdx.y);
    const int lane16 = threadIdx.x & 15;
    const int nwin = a.cur.nwx * a.cur.nwy;
    for (int w = blockIdx.x * 16 + (threadIdx.x >> 4); w < nwin; w += gridDim.x * 16) {
        const int wy = w / a.cur.nwx, wx = w - wy * a.cur.nwx;
        const WinConst wc = load_win_const(g, a, wx, wy, a.axis == 1);
        const int searched0 = a.axis ? wc.oy : wc.ox;
        Best b{0xFFFFFFFFu, 16};
        uint32_t mine = 0;
        if (lane16 < a.R) {
            const int cand = (int)(int16_t)(searched0 + rel_offset(lane16, a.R));
            mine = (a.sums[w * 16 + lane16] << a.delta_scalar) +
                   wc.npix * window_bias(cand, a.use_neighbors, a.axis ? wc.nby : wc.nbx, a.neighbor_scalar);
            b.sum = mine; b.cz = lane16;
        }
        const int lane = (int)(threadIdx.x & 63);
        best_xor<1>(b, lane); best_xor<2>(b, lane); best_xor<4>(b, lane); best_xor<8>(b, lane);
        if (lane16 == 0) {
            int16_t* t = a.axis ? a.cur.ty : a.cur.tx;
            t[w] = (int16_t)(searched0 + rel_offset(b.cz, a.R));
        }
        if (a.capture_delta && w == 0 && lane16 == (a.R >> 1) - 1) *a.total_delta = mine / a.delta_divisor;
    }
}

// m_offsetArray view of the last level: int16 [2][lh][lw]
__global__ void expand_offsets_kernel(const Geom g, const FlowLevel L, int16_t* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.lw || y >= g.lh) return;
    const size_t p = (size_t)y * g.lw + x, N = (size_t)g.lw * g.lh;
    out[p] = L.tx ? (int16_t)table_at(L.tx, L, x, y) : (int16_t)0;
    out[N + p] = L.ty ? (int16_t)table_at(L.ty, L, x, y) : (int16_t)0;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
PhaseLayout make_phase_layout(const Geom& g, int max_iterations) {
    PhaseLayout pl{};
    pl.rs = g.rs;
    pl.nph = 1 << g.rs;
    pl.nph2 = pl.nph > 1 ? pl.nph / 2 : 1;
    const int reach = (max_iterations + 1) * 64 + 8;   // |offset| <= iterations * (R/2)^2, + one candidate, R <= 16
    pl.mx = (((reach >> g.rs) + 2 + 3) / 4) * 4;        // multiple of 4: margin groups line up with the 4-column groups
    pl.lwp = ((g.lw + 2 * pl.mx + 4 + 31) / 32) * 32;   // a strip may start at column lw + mx; rows start on a 128-byte line
    pl.bytes = (size_t)g.H * pl.nph2 * pl.lwp * sizeof(uint32_t);
    return pl;
}

template <typename E>
static bool launch_prep_fast(const Geom& g, const PhaseLayout& pl, const PrepBatch& b, hipStream_t stream) {
    const int lw = g.W >> g.rs;
    const size_t row_bytes = (size_t)g.in_stride * sizeof(E);
    if ((lw << g.rs) != g.W || lw != g.lw || (lw & 3) || pl.mx > lw || (pl.mx & 3) || (row_bytes & 15) || (pl.lwp & 3) || g.rs > 4 || (g.H & 1))
        return false;
    for (int i = 0; i < b.n; i++) if (((uintptr_t)b.frame[i]) & 15) return false;
    const dim3 grd((lw / 4 + 127) / 128, g.H / 2, b.n);
    const bool nt = b.n > 1;     // batches: non-temporal plane stores (hf_phase_plane.h); a single context's plane stays in the caches for its chain
    switch (g.rs) {
        case 0: if (nt) HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 0, true>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); else HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 0, false>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); break;
        case 1: if (nt) HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 1, true>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); else HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 1, false>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); break;
        case 2: if (nt) HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 2, true>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); else HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 2, false>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); break;
        case 3: if (nt) HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 3, true>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); else HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 3, false>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); break;
        default: if (nt) HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 4, true>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); else HF_LAUNCH("plane", (prep_phase_fast_kernel<E, 4, false>), grd, dim3(128), 0, stream, b, g.H, g.W, g.in_stride, pl); break;
    }
    return true;
}

void launch_prep_frames(const Geom& g, const PhaseLayout& pl, const PrepBatch& b, hipStream_t stream) {
    if (g.hdr ? launch_prep_fast<uint16_t>(g, pl, b, stream) : launch_prep_fast<uint8_t>(g, pl, b, stream)) return;
    const size_t smem = 2 * (size_t)((g.W + 15) / 16) * 16;
    const dim3 grd(g.H, b.n);
    if (g.hdr) HF_LAUNCH("plane", (prep_phase_kernel<uint16_t>), grd, dim3(256), smem, stream, b, g.H, g.W, g.in_stride, pl);
    else HF_LAUNCH("plane", (prep_phase_kernel<uint8_t>), grd, dim3(256), smem, stream, b, g.H, g.W, g.in_stride, pl);
}

// Grid samples only: element (cy << rs, phase pair 0, column j) of the plane = what load_strip reads of the newer frame.
template <typename E>
__global__ __launch_bounds__(256) void prep_grid_kernel(const PrepBatch batch, int H, int S, PhaseLayout pl, int lw) {
    const E* __restrict__ f = (const E*)batch.frame[blockIdx.z];
    uint32_t* __restrict__ pp = batch.pp[blockIdx.z];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= lw) return;
    const int sy = (int)blockIdx.y << pl.rs, x = j << pl.rs;     // rs >= 1: x is even and x + 1 lies inside the row
    HF_DBG_CHECK(sy < H && x + 1 < S && ((size_t)sy * pl.nph2 * pl.lwp + pl.mx + j + 1) * 4 <= pl.bytes, 107);
    const E* __restrict__ yr = f + (size_t)sy * S + x;
    const E* __restrict__ cr = f + (size_t)H * S + (size_t)(sy >> 1) * S + x;
    // (non-temporal loads: a 128-byte line is fetched for 4 + 4 useful bytes and nothing reads the frame again before the next period's warp
    //  launch, 47 other pair streams later -- the lines should not push the chain's plane rows out of L2: +1.3-2.8 % on the 2160p pipeline)
    typedef E e2 __attribute__((ext_vector_type(2)));
    const e2 y2 = __builtin_nontemporal_load((const e2*)yr), c2 = __builtin_nontemporal_load((const e2*)cr);
    pp[(size_t)sy * pl.nph2 * pl.lwp + pl.mx + j] = pack_element(top8<E>(y2.x), top8<E>(y2.y), top8<E>(c2.x), top8<E>(c2.y));
}

void launch_prep_grid(const Geom& g, const PhaseLayout& pl, const PrepBatch& b, hipStream_t stream) {
    const dim3 grd((g.lw + 255) / 256, g.lh, b.n);
    if (g.hdr) HF_LAUNCH("grid_samples", (prep_grid_kernel<uint16_t>), grd, dim3(256), 0, stream, b, g.H, g.in_stride, pl, g.lw);
    else HF_LAUNCH("grid_samples", (prep_grid_kernel<uint8_t>), grd, dim3(256), 0, stream, b, g.H, g.in_stride, pl, g.lw);
}

void launch_prep_frame(const Geom& g, const PhaseLayout& pl, const void* frame, uint32_t* pp, hipStream_t stream) {
    PrepBatch b{};
    b.n = 1; b.frame[0] = frame; b.pp[0] = pp;
    launch_prep_frames(g, pl, b, stream);
}

// Batches up to this size run the two finest levels with one row per lane (MapRow).  Chain alone, us per batched chain with a block /
// a row per lane: 1 pair 79.5 / 71.3, 2 pairs 94.6 / 86.8, 4 pairs 122.4 / 119.2, 8 pairs 169.3 / 173.8.
constexpr int kRowPerLaneMaxBatch = 4;
#ifndef HF_LEVEL32_ONE_WAVE_MIN_BATCH
#define HF_LEVEL32_ONE_WAVE_MIN_BATCH 4
#endif
constexpr int kLevel32OneWaveMinBatch = HF_LEVEL32_ONE_WAVE_MIN_BATCH;   // batches from this size on: level 32 as one wave per window
void launch_flow_level_small(const Geom& g, const FlowBatch& b, hipStream_t stream) {
    const int ws = b.s[0].cur.window;
    const bool rows1 = b.n <= kRowPerLaneMaxBatch && ws <= 4;
    const int tw = rows1 && ws == 2 ? MapRow<2>::TW : 32;
    const int tiles_x = (g.lw + tw - 1) / tw, tiles_y = (g.lh + 31) / 32;   // (32 x 32 tiles at every level but MapRow<2>: 16 x 32)
    FlowBatchArgs kb = pack_batch(b, tiles_x, tiles_y);
    // dynamic LDS: the candidate rows of the Y step (full tiles only exist at the full search radius)
    const size_t lds = b.s[0].R != 16 ? 0 : ws == 32 ? ystage_bytes<32, 8, 4>(g.rs) : ws == 16 ? ystage_bytes<16, 4, 1>(g.rs) : ws == 8 ? win8_stage_bytes(g.rs) : 0;
    // windows <= 16 never span waves: one-wave workgroups (SPLIT), see flow_level_small_kernel
    const dim3 grd(xcd_grid(tiles_x, tiles_y, 1, b.n));
    auto split = [&](int waves) { return dim3(xcd_grid(tiles_x, tiles_y, waves, b.n)); };
    kb.common.y_rows_lds = lds != 0;
    const bool tabk = b.s[0].sadtab && b.s[0].R == 16 && (b.s[0].sad_read || b.s[0].sad_write);
#define HF_LEVEL(NAME, WS_, SPLIT_, ROWS1_, GRID, BLOCK, LDS)                                                                            \
    do {                                                                                                                                 \
        if (tabk) HF_LAUNCH(NAME, (flow_level_small_kernel<WS_, SPLIT_, ROWS1_, true>), GRID, BLOCK, LDS, stream, g, kb);                \
        else HF_LAUNCH(NAME, (flow_level_small_kernel<WS_, SPLIT_, ROWS1_, false>), GRID, BLOCK, LDS, stream, g, kb);                    \
    } while (0)
    switch (ws) {
        case 32:
            if (b.n >= kLevel32OneWaveMinBatch) {      // one wave per window (flow_level32_wave_kernel)
                const size_t lds1 = b.s[0].R != 16 ? 0 : ystage_bytes<32, 8, 1>(g.rs);
                kb.common.y_rows_lds = lds1 != 0;
                if (tabk) HF_LAUNCH("level_32", (flow_level32_wave_kernel<true>), grd, dim3(64), lds1, stream, g, kb);
                else HF_LAUNCH("level_32", (flow_level32_wave_kernel<false>), grd, dim3(64), lds1, stream, g, kb);
                break;
            }
            HF_LEVEL("level_32", 32, false, false, grd, dim3(256), lds);
            break;
        case 16: HF_LEVEL("level_16", 16, true, false, split(Map<16>::WAVES), dim3(64), lds); break;
        case 8: HF_LEVEL("level_8", 8, true, false, split(Map<8>::WAVES), dim3(64), lds); break;
        case 4:
            if (rows1) HF_LEVEL("level_4", 4, true, true, split(MapRow<4>::WAVES), dim3(64), 0);
            else HF_LEVEL("level_4", 4, true, false, split(Map<4>::WAVES), dim3(64), 0);
            break;
        default:
            if (rows1) HF_LEVEL("level_2", 2, true, true, split(MapRow<2>::WAVES), dim3(64), 0);
            else HF_LEVEL("level_2", 2, true, false, split(Map<2>::WAVES), dim3(64), 0);
            break;
    }
#undef HF_LEVEL
}

// Waves per workgroup.  A chain alone (2160p HDR, 16 pairs): 4 waves 214 us, 1 wave 224 us (more atomics, and the Y launch's candidate rows
// come out of L2 instead of LDS).  Inside a throughput pipeline the other queues' kernels hold most of every CU and a single wave finds
// room sooner: same-box A-B with one-wave workgroups 1080p SDR + 1.4-3 %, 2160p SDR + 2.5 %, 64 pairs + 0.9 %, 1080p HDR + 0.3 %, 2160p HDR
// (bandwidth-bound) +- 0; 360p (rs = 1) - 1.5 %.  A throughput driver's batches at rs >= 2 take one wave.  (Measured, round 6; before level 32
// became a one-wave launch too, 2160p HDR lost 2 % with them; a 16 x 16 one-wave Y tile with staged rows was no better than the plain one.)
#ifndef HF_BIG_ONE_WAVE_MIN_BATCH
#define HF_BIG_ONE_WAVE_MIN_BATCH 4
#endif
constexpr int kBigWavesPerBlock = 4, kBigOneWaveMinBatch = HF_BIG_ONE_WAVE_MIN_BATCH, kBigOneWaveMinRs = 2;
void launch_flow_big_partial(const Geom& g, const FlowBatch& b, hipStream_t stream) {
    const bool y = b.s[0].axis == 1;
    const char* name = y ? "large_windows_y" : "large_windows_x";
    const int wpb = b.n >= kBigOneWaveMinBatch && g.rs >= kBigOneWaveMinRs ? 1 : kBigWavesPerBlock;
    const int tiles_x = (g.lw + 63) / 64, tiles_y = (g.lh + 4 * wpb - 1) / (4 * wpb);
    const dim3 grd(xcd_grid(tiles_x, tiles_y, 1, b.n));
    const size_t lds = y && b.s[0].R == 16 && wpb == 4 ? ystage_bytes<16, 16, 4>(g.rs) : 0;   // Y launches: candidate rows
    FlowBatchArgs kb = pack_batch(b, tiles_x, tiles_y);
    kb.common.y_rows_lds = lds != 0;
    if (wpb == 1) HF_LAUNCH(name, (flow_big_partial_kernel<1>), grd, dim3(64), lds, stream, g, kb);
    else HF_LAUNCH(name, (flow_big_partial_kernel<kBigWavesPerBlock>), grd, dim3(64 * kBigWavesPerBlock), lds, stream, g, kb);
}

void launch_flow_big_argmin(const Geom& g, const FlowBatch& b, hipStream_t stream) {
    const int nwin = b.s[0].cur.nwx * b.s[0].cur.nwy;
    HF_LAUNCH("large_windows_argmin", flow_big_argmin_kernel, dim3((nwin + 15) / 16, b.n), dim3(256), 0, stream, g, pack_batch(b, 1, 1));
}

bool dbg_bounds_read_flow(unsigned out[5], bool reset) {
#ifdef HF_DEBUG_BOUNDS
    unsigned rec[5] = {0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(rec, HIP_SYMBOL(g_dbg_bounds), sizeof(rec)) != hipSuccess) return false;
    if (rec[0] && !out[0]) for (int i = 1; i < 5; i++) out[i] = rec[i];
    out[0] += rec[0];
    if (reset) { const unsigned zero[5] = {0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_bounds), zero, sizeof(zero)); }
    return true;
#else
    (void)out; (void)reset;
    return false;
#endif
}

void launch_expand_offsets(const Geom& g, const FlowLevel& last, int16_t* out, hipStream_t stream) {
    const dim3 grd((g.lw + 63) / 64, (g.lh + 3) / 4);
    expand_offsets_kernel<<<grd, 256, 0, stream>>>(g, last, out);
}

}  // namespace hf
