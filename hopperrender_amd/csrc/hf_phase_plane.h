// hf_phase_plane.h -- device code that builds the phase plane of a frame (the re-laid top-8-bit copy of a frame that
// replaces the strided candidate sampling of calcDeltaSumsKernelSDR.h:78-100).  Shared by the stand-alone plane kernel (hf_flow.hip)
// and the fused period warp, whose launch carries plane-building workgroups for the frame it reads anyway (hf_kernels.hip).
#pragma once
#include "hf_kernels.h"

namespace hf {
namespace {

template <typename E> __device__ __forceinline__ unsigned top8(E v);
template <> __device__ __forceinline__ unsigned top8<uint8_t>(uint8_t v) { return v; }
template <> __device__ __forceinline__ unsigned top8<uint16_t>(uint16_t v) { return (unsigned)(v >> 8); }  // calcDeltaSumsKernelHDR.h:98

__device__ __forceinline__ uint32_t pack_element(uint32_t ya, uint32_t yb, uint32_t u, uint32_t v) {
    return ya | (yb << 8) | (u << 16) | (v << 24);
}


// Fast plane build (no LDS), one task: the 4 << RS consecutive elements behind 4 consecutive grid columns (4 t .. 4 t + 3) of the
// luma rows 2m, 2m + 1 and of their chroma row m -> one 16-byte store (4 columns) per phase pair and luma row.  A wave of tasks
// with consecutive t reads 64 x 16..64 contiguous bytes per row and writes 1 KB per phase row.
// The mirrored margins need no extra loads:
//     PP[ph2][-1-k] = swap(PP[nph2-1-ph2][k])        PP[ph2][lw+k] = swap(PP[nph2-1-ph2][lw-1-k])
// (swap = the two luma bytes exchanged; rs = 0 has one luma byte per element and no swap), because reflecting
// x -> -x-1 (or 2W-x-1) maps phase ph of column j to phase nph-1-ph of column -j-1 and keeps the chroma pair, so a
// task whose columns lie within `mx` of an edge also stores its elements, columns reversed, into the margin.
// Requires W == lw << RS, lw % 4 == 0, mx <= lw and 16-byte aligned rows (else: prep_phase_kernel).
// NZ = 2: both luma rows of chroma row m (z0 = 0; the stand-alone kernel); NZ = 1: luma row 2m + z0 only (the fused warp launch,
// half the registers per thread).
// NT: non-temporal plane stores (below) -- for the launches of a throughput pipeline; a lone stream keeps the default policy (its chain reads
// the plane right behind the build, with nothing else competing for the caches).
template <typename E, int RS, int NZ, bool NT = true>
__device__ __forceinline__ void plane_fast_task(const E* __restrict__ f, uint32_t* __restrict__ pp, const int H, const int W, const int S,
                                                const PhaseLayout& pl, const int m, const int t, const int z0) {
    constexpr int NPH = 1 << RS, NE = 4 << RS;               // phases, elements per task and row
    constexpr int NPH2 = NPH > 1 ? NPH / 2 : 1;
    const int lw = W >> RS;
    if (4 * t >= lw) return;
    HF_DBG_CHECK(2 * m + z0 + NZ <= H && 2 * m < H && (size_t)(t + 1) * NE <= (size_t)S && ((size_t)(2 * m + z0 + NZ) * NPH2 * pl.lwp) * 4 <= pl.bytes, 120);
    __attribute__((aligned(16))) E e[NZ + 1][NE];            // luma row(s) 2m + z0 .., chroma row m (last)
#pragma unroll
    for (int z = 0; z <= NZ; z++) {
        const E* __restrict__ src = (z < NZ ? f + (size_t)(2 * m + z0 + z) * S : f + (size_t)H * S + (size_t)m * S) + (size_t)t * NE;
        if (NE * sizeof(E) >= 16) {
#pragma unroll
            for (int i = 0; i < NE * (int)sizeof(E) / 16; i++) ((uint4*)e[z])[i] = ((const uint4*)src)[i];
        } else {                                             // RS = 0..1 with 8-bit elements: 4 or 8 bytes per thread
#pragma unroll
            for (int i = 0; i < NE; i++) e[z][i] = src[i];
        }
    }
    const int j0 = 4 * t;                                    // first column of this thread
    const int jl = pl.mx - 4 - j0;                           // plane index of the mirrored group in the left margin
    const int jr = pl.mx + 2 * lw - 4 - j0;                  // ... and in the right margin
    const bool left = j0 + 4 <= pl.mx, right = j0 >= lw - pl.mx;
#pragma unroll
    for (int z = 0; z < NZ; z++) {
        uint32_t* __restrict__ base = pp + (size_t)(2 * m + z0 + z) * NPH2 * pl.lwp;
#pragma unroll
        for (int p2 = 0; p2 < NPH2; p2++) {
            uint32_t el[4], sw[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int x = RS > 0 ? c * NPH + 2 * p2 : c;     // element index of the column's sample inside e[]
                const int xc = x & ~1;
                const uint32_t ya = top8<E>(e[z][x]), yb = RS > 0 ? top8<E>(e[z][x + 1]) : 0u;
                const uint32_t u = top8<E>(e[NZ][xc]), v = top8<E>(e[NZ][xc + 1]);
                el[c] = pack_element(ya, yb, u, v);
                sw[c] = RS > 0 ? pack_element(yb, ya, u, v) : el[c];
            }
            // Non-temporal stores: the chain reads the plane one to twelve launches later, by which time a batch's planes (12 x 22 MB) have
            // long left the 4 MB L2s whatever the policy -- but written with the default policy they push out what the launch still needs
            // (source rows shared by neighbouring tiles and by these workgroups, the flow tables).  +2-4 % frames/s on the 2160p HDR
            // pipeline, alternating on one box (tools/attic/r05); the frame LOADS above must keep the default policy (non-temporal: -7 %).
            // (NT = false: a single context's plane kernel -- the lone stream's chain, which follows at once, then finds the plane in the caches:
            //  85 -> 82 us per flow calculation on a new frame.)
            typedef unsigned pp_v4 __attribute__((ext_vector_type(4)));
            auto put = [](uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
                const pp_v4 v = {a, b, c, d};
                if constexpr (NT) __builtin_nontemporal_store(v, (pp_v4*)dst); else *(pp_v4*)dst = v;
            };
            put(base + (size_t)p2 * pl.lwp + pl.mx + j0, el[0], el[1], el[2], el[3]);
            uint32_t* mrow = base + (size_t)(NPH2 - 1 - p2) * pl.lwp;
            if (left) put(mrow + jl, sw[3], sw[2], sw[1], sw[0]);
            if (right) put(mrow + jr, sw[3], sw[2], sw[1], sw[0]);
        }
    }
}

}  // namespace
}  // namespace hf
