// hopperrender_amd/csrc/hf_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the
// OpticalFlowCalc hot path.  Wave = 64 lanes everywhere.  Build with -ffp-contract=off: the
// only fused operations are the ones written explicitly (see "fp32 flavour" below).
//
// What each kernel replaces (reference HopperRender/):
//   (the refinement chain -- calcDeltaSums / determineLowestLayer / adjustOffsetArray -- is hf_flow.hip)
//   blur_flow_kernel     - blurFlowKernelSDR.h:17-92, separable through LDS, runtime radius; the reference's radius on a chain that ended at
//                          2 x 2 windows: the 8 x 8 box sum as four reads of 4 x 4 WINDOW sums.
//   warp_kernel          - warpFrameKernel{SDR,HDR}.h:116-184 (all 7 modes), both planes, one launch.
//   copy_kernel          - copyFrameKernel{SDR,HDR}.h:12-25, both planes, one launch.
//
// fp32 flavour: blend/levels reproduce the reference AS IT RUNS ON gfx950 through AMD OpenCL
// (measured, tests/golden/levels_ramp.npz): "a*u + b*t" is contracted to fma(a, u, b*t); the
// division in apply_levels* is x * v_rcp_f32(y); "q*max + mid" is fma(q, max, mid).
#include "hf_kernels.h"
#include "hf_phase_plane.h"
#include <type_traits>

#include <hip/hip_ext.h>

namespace hf {

namespace {

template <typename E> struct ElemTraits;
template <> struct ElemTraits<uint8_t> {
    static constexpr bool hdr = false;
    static constexpr float maxv = 255.0f;
    static constexpr float mid = 128.0f;
    static constexpr unsigned midu = 128u;
    __device__ static __forceinline__ unsigned top8(uint8_t v) { return v; }
};
template <> struct ElemTraits<uint16_t> {
    static constexpr bool hdr = true;
    static constexpr float maxv = 65535.0f;
    static constexpr float mid = 32768.0f;
    static constexpr unsigned midu = 32768u;
    __device__ static __forceinline__ unsigned top8(uint16_t v) { return (unsigned)(v >> 8); }  // calcDeltaSumsKernelHDR.h:98
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
// 16-byte store of an output frame: nothing on the GPU reads output frames back, so they should not displace the
// source frames / phase planes in L2 and the Infinity Cache (non-temporal hint; measured on the fused 2160p HDR period
// with rotating buffers: 65.7 -> 54.5 us, default bench 48.2k -> 51.2k frames/s)
typedef unsigned nt_uint4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_streaming(void* dst, const void* src16) {
    __builtin_nontemporal_store(*(const nt_uint4*)src16, (nt_uint4*)dst);
}
__device__ __forceinline__ unsigned absdiff(unsigned a, unsigned b) { return a > b ? a - b : b - a; }

// sgn(d)*d*d, d = layer - R/2 (calcDeltaSumsKernelSDR.h:70-74)
__device__ __forceinline__ int rel_offset(int layer, int R) {
    const int d = layer - (R >> 1);
    return d > 0 ? d * d : -(d * d);
}

// ------------------------------------------------------------------------------------------
// blur
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int mirror_flow(int pos, int dim) {  // blurFlowKernelSDR.h:7-14 (+ safety clamp)
    if (pos >= dim) pos = 2 * dim - pos - 1; else if (pos < 0) pos = -pos - 1;
    return clampi(pos, 0, dim - 1);
}

// One workgroup = TS x TS outputs, both planes: the offsets of the (TS + 2r)^2 neighbourhood go through LDS once as
// packed x | y << 16 words, then a horizontal and a vertical pass of 2r taps each (blurFlowKernelSDR.h:79-91 sums
// the same (2r)^2 taps in one double loop; integer sums are order-independent).  The kernel also emits the packed
// copy of the blurred flow that the warp kernels read with one load, and re-zeroes the window sums of the chain.
// TS = 32 with the radius fixed at compile time (RFIX = 4, the reference's) is the chain's launch: a quarter of the workgroups of the
// 16 x 16 version (one round of waves instead of four for a batch of 16), 1.56 instead of 2.25 gathered offsets per output, unrolled taps;
// TS = 16 with a runtime radius (RFIX = 0) serves every other radius (up to 64: 101 KB of LDS).
__device__ __forceinline__ int div_trunc(int s, int d, int log2d) {   // C division (truncation toward zero), :89-90
    return log2d >= 0 ? (s + ((s >> 31) & (d - 1))) >> log2d : s / d;
}
template <int TS, int RFIX>
__global__ __launch_bounds__(256) void blur_flow_kernel(const BlurBatch batch, int lw, int lh, int r_arg, int zero_count) {
    const BlurItem& it = batch.s[blockIdx.z];   // blockIdx.z: pair of the batch
    const FlowLevel& L = it.last;
    int16_t* __restrict__ blurred = it.blurred;
    uint32_t* __restrict__ packed = it.packed;
    uint32_t* __restrict__ zero = it.zero;
    const int r = RFIX ? RFIX : r_arg;
    if (it.still_count && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // the chain's content hint: out to the host, cleared for the next chain
        *it.still_out = *it.still_count;
        *it.still_count = 0u;
    }
    if (zero) {   // the refinement steps are done with the window sums: clear them for the next chain
        const int nthreads = gridDim.x * gridDim.y * 256;
        for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero_count; i += nthreads) zero[i] = 0u;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (TS == 32 && RFIX == 4) {
        // A chain that ended at 2 x 2 windows on a grid of even size (the usual case): the 8 taps of an output are pixels X - 4 .. X + 3 = four
        // whole windows for even X, half + three + half windows for odd X, so with S4[i] = v[i] + .. + v[i + 3] over WINDOWS the row sum is
        // S4[a] + S4[a + (X & 1)], a = (X - 4 - (X & 1)) / 2, and the 8 x 8 sum four reads of W44 = the 4 x 4 window sums: 20 x 20 gathered
        // offsets instead of 40 x 40, 4 + 4 LDS reads per output instead of 8 + 16.  Integer sums: the same value as the tap loops below.
        // (Even sizes: the reflection of blurFlowKernelSDR.h:7-14 then maps pixel pairs onto pixel pairs, i.e. windows onto windows.)
        const int X0 = blockIdx.x * 32, Y0 = blockIdx.y * 32;
        if (L.tx && L.ty && L.log2w == 1 && !(lw & 1) && !(lh & 1) && lw >= 64 && lh >= 64 && L.nwx * 2 == lw && L.nwy * 2 == lh) {   // kernel-uniform
            constexpr int NW = 20, NS = 17;                   // windows per tile edge, S4 / W44 values per edge
            uint32_t* v = (uint32_t*)smem;                    // [NW][NW] packed offsets
            int* hsx = (int*)(v + NW * NW);                   // [NW][NS] horizontal 4-window sums
            int* hsy = hsx + NW * NS;
            int* wx = hsy + NW * NS;                          // [NS][NS] 4 x 4 window sums
            int* wy = wx + NS * NS;
            const int tid = threadIdx.x;
            const int wa0 = X0 / 2 - 2, wb0 = Y0 / 2 - 2;
            for (int i = tid; i < NW * NW; i += 256) {
                const int j = i / NW, k = i - j * NW;
                int wa = wa0 + k, wb = wb0 + j;
                wa = wa < 0 ? -1 - wa : wa >= L.nwx ? 2 * L.nwx - 1 - wa : wa;
                wb = wb < 0 ? -1 - wb : wb >= L.nwy ? 2 * L.nwy - 1 - wb : wb;
                const int w = wb * L.nwx + wa;
                HF_DBG_CHECK(wb >= 0 && wb < L.nwy && wa >= 0 && wa < L.nwx, 201);
                v[i] = (uint32_t)(uint16_t)L.tx[w] | ((uint32_t)(uint16_t)L.ty[w] << 16);
            }
            __syncthreads();
            for (int i = tid; i < NW * NS; i += 256) {
                const int j = i / NS, k = i - j * NS;
                int sx = 0, sy = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) { const uint32_t w = v[j * NW + k + t]; sx += (int)(int16_t)(w & 0xFFFFu); sy += (int)w >> 16; }
                hsx[i] = sx; hsy[i] = sy;
            }
            __syncthreads();
            for (int i = tid; i < NS * NS; i += 256) {
                const int j = i / NS, k = i - j * NS;
                int sx = 0, sy = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) { sx += hsx[(j + t) * NS + k]; sy += hsy[(j + t) * NS + k]; }
                wx[i] = sx; wy[i] = sy;
            }
            __syncthreads();
            const int tx = tid & 31, ty = tid >> 5;           // 32 x 8 threads, four output rows each
            const int ax = (tx - (tx & 1)) / 2, dx = tx & 1;  // (X0 is even: the parity of X is tx's)
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int cy = ty + 8 * o, ay = (cy - (cy & 1)) / 2, dy = cy & 1;
                const int i00 = ay * NS + ax, i01 = i00 + dx, i10 = i00 + dy * NS, i11 = i10 + dx;
                const int sx = wx[i00] + wx[i01] + wx[i10] + wx[i11], sy = wy[i00] + wy[i01] + wy[i10] + wy[i11];
                const int rx = (int)(int16_t)div_trunc(sx, 64, 6), ry = (int)(int16_t)div_trunc(sy, 64, 6);
                if (X0 + tx < lw && Y0 + cy < lh) {
                    const size_t q = (size_t)(Y0 + cy) * lw + X0 + tx;
                    blurred[q] = (int16_t)rx;
                    blurred[(size_t)lw * lh + q] = (int16_t)ry;
                    packed[q] = ((uint32_t)rx & 0xFFFFu) | ((uint32_t)ry << 16);
                }
            }
            return;
        }
    }
    if constexpr (TS == 32 && RFIX == 0) {
        // The window-sum form at a run-time EVEN radius (launch_blur_flow starts this instantiation only where it applies: a chain that ended
        // at 2 x 2 windows on a grid of even size): the 2r taps of an output are r whole windows for even X, half + (r - 1) + half for odd X, so
        // with S_r[i] = v[i] + .. + v[i + r - 1] over WINDOWS the row sum is S_r[a] + S_r[a + (X & 1)], a = (X - r - (X & 1)) / 2, and the
        // 2r x 2r sum four reads of the r x r window sums.  (16 + r)^2 gathered offsets per 32 x 32 outputs (radius 32: 2.25 per output where
        // the pixel form gathers 25), running sums along rows and columns, 18 KB of LDS instead of 36: BASELINE config 5's blur.
        const int X0 = blockIdx.x * 32, Y0 = blockIdx.y * 32;
        const int NW = 16 + r;                                // windows per tile edge
        constexpr int NS = 17;                                // S_r / window-sum values per edge
        uint32_t* v = (uint32_t*)smem;                        // [NW][NW] packed offsets
        int* hsx = (int*)(v + NW * NW);                       // [NW][NS] horizontal r-window sums
        int* hsy = hsx + NW * NS;
        int* wx = hsy + NW * NS;                              // [NS][NS] r x r window sums
        int* wy = wx + NS * NS;
        const int tid = threadIdx.x;
        const int wa0 = X0 / 2 - r / 2, wb0 = Y0 / 2 - r / 2;
        HF_DBG_CHECK(L.tx && L.ty && L.log2w == 1 && !(r & 1) && L.nwx * 2 == lw && L.nwy * 2 == lh, 215);
        for (int i = tid; i < NW * NW; i += 256) {
            const int j = i / NW, k = i - j * NW;
            int wa = wa0 + k, wb = wb0 + j;
            wa = wa < 0 ? -1 - wa : wa >= L.nwx ? 2 * L.nwx - 1 - wa : wa;
            wb = wb < 0 ? -1 - wb : wb >= L.nwy ? 2 * L.nwy - 1 - wb : wb;
            const int w = wb * L.nwx + wa;
            HF_DBG_CHECK(wb >= 0 && wb < L.nwy && wa >= 0 && wa < L.nwx, 201);
            v[i] = (uint32_t)(uint16_t)L.tx[w] | ((uint32_t)(uint16_t)L.ty[w] << 16);
        }
        __syncthreads();
        for (int row = tid; row < NW; row += 256) {           // a lane walks a row with a running sum of r windows
            const uint32_t* p = v + row * NW;
            int sx = 0, sy = 0;
            for (int t = 0; t < r; t++) { const uint32_t w = p[t]; sx += (int)(int16_t)(w & 0xFFFFu); sy += (int)w >> 16; }
            hsx[row * NS] = sx; hsy[row * NS] = sy;
            for (int k = 1; k < NS; k++) {
                const uint32_t win = p[k + r - 1], wout = p[k - 1];
                sx += (int)(int16_t)(win & 0xFFFFu) - (int)(int16_t)(wout & 0xFFFFu);
                sy += ((int)win >> 16) - ((int)wout >> 16);
                hsx[row * NS + k] = sx; hsy[row * NS + k] = sy;
            }
        }
        __syncthreads();
        for (int col = tid; col < 2 * NS; col += 256) {       // ... and a column of the row sums (x and y: a lane each)
            const int* h = col < NS ? hsx + col : hsy + col - NS;
            int* o = col < NS ? wx + col : wy + col - NS;
            int sum = 0;
            for (int t = 0; t < r; t++) sum += h[t * NS];
            o[0] = sum;
            for (int j = 1; j < NS; j++) { sum += h[(j + r - 1) * NS] - h[(j - 1) * NS]; o[j * NS] = sum; }
        }
        __syncthreads();
        const int d = 4 * r * r, log2d = (r & (r - 1)) == 0 ? 2 + 2 * (31 - __builtin_clz(r)) : -1;
        const int tx = tid & 31, ty = tid >> 5;               // 32 x 8 threads, four output rows each
        const int ax = (tx - (tx & 1)) / 2, dx = tx & 1;      // (X0 is even: the parity of X is tx's)
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const int cy = ty + 8 * o, ay = (cy - (cy & 1)) / 2, dy = cy & 1;
            const int i00 = ay * NS + ax, i01 = i00 + dx, i10 = i00 + dy * NS, i11 = i10 + dx;
            const int sx = wx[i00] + wx[i01] + wx[i10] + wx[i11], sy = wy[i00] + wy[i01] + wy[i10] + wy[i11];
            const int rx = (int)(int16_t)div_trunc(sx, d, log2d), ry = (int)(int16_t)div_trunc(sy, d, log2d);
            if (X0 + tx < lw && Y0 + cy < lh) {
                const size_t q = (size_t)(Y0 + cy) * lw + X0 + tx;
                blurred[q] = (int16_t)rx;
                blurred[(size_t)lw * lh + q] = (int16_t)ry;
                packed[q] = ((uint32_t)rx & 0xFFFFu) | ((uint32_t)ry << 16);
            }
        }
        return;
    }
    const int T = TS + 2 * r, TP = T + 1;         // tile edge; row pitch (odd: the row-per-lane pass below walks the banks)
    uint32_t* tile = (uint32_t*)smem;             // [T][TP] packed offsets
    int* hx = (int*)(tile + T * TP);              // [T][TS] horizontal sums of x
    int* hy = hx + T * TS;                        // [T][TS] ... of y
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * TS - r, y0 = blockIdx.y * TS - r;
    for (int py = ty; py < T; py += 16) {         // offsets are stored per window of the last level
        const int wrow = (mirror_flow(y0 + py, lh) >> L.log2w) * L.nwx;
        for (int px = tx; px < T; px += 16) {
            const int w = wrow + (mirror_flow(x0 + px, lw) >> L.log2w);
            HF_DBG_CHECK(w >= 0 && w < L.nwx * L.nwy && py * TP + px < T * TP, 200);
            const uint32_t ox = L.tx ? (uint32_t)(uint16_t)L.tx[w] : 0u, oy = L.ty ? (uint32_t)(uint16_t)L.ty[w] : 0u;
            tile[py * TP + px] = ox | (oy << 16);
        }
    }
    __syncthreads();
    if constexpr (RFIX) {
        for (int idx = tid; idx < T * TS; idx += 256) {   // taps -r .. r-1 (blurFlowKernelSDR.h:82-83)
            const int row = idx / TS, col = idx - row * TS;    // (TS is a power of two)
            const uint32_t* p = tile + row * TP + col;
            int sx = 0, sy = 0;
#pragma unroll
            for (int k = 0; k < 2 * RFIX; k++) { const uint32_t w = p[k]; sx += (int)(int16_t)(w & 0xFFFFu); sy += (int)w >> 16; }
            hx[idx] = sx;
            hy[idx] = sy;
        }
    } else {
        // Run-time radius: a lane walks HALF A ROW with a running sum -- 2r taps for the first output, then one tap in and one out per
        // output -- instead of 2r taps for every output (radius 32: 78 tap reads per 8 outputs where the tap loops read 512; the launch
        // was bound by the unpack-and-add of those taps: 104 us per 12 pairs alone, 365 us inside the pipeline of BASELINE config 5).
        // The same integers: sums of int16 offsets in 32 bits, whatever the order.
        constexpr int HALF = TS / 2;
        for (int task = tid; task < 2 * T; task += 256) {
            const int row = task >> 1, c0 = (task & 1) * HALF;
            const uint32_t* p = tile + row * TP + c0;
            int sx = 0, sy = 0;
            for (int k = 0; k < 2 * r; k++) { const uint32_t w = p[k]; sx += (int)(int16_t)(w & 0xFFFFu); sy += (int)w >> 16; }
            hx[row * TS + c0] = sx;
            hy[row * TS + c0] = sy;
#pragma unroll
            for (int c = 1; c < HALF; c++) {
                const uint32_t win = p[c + 2 * r - 1], wout = p[c - 1];
                sx += (int)(int16_t)(win & 0xFFFFu) - (int)(int16_t)(wout & 0xFFFFu);
                sy += ((int)win >> 16) - ((int)wout >> 16);
                hx[row * TS + c0 + c] = sx;
                hy[row * TS + c0 + c] = sy;
            }
        }
    }
    __syncthreads();
    const int d = 4 * r * r, log2d = (r & (r - 1)) == 0 ? 2 + 2 * (31 - __builtin_clz(r)) : -1;
#pragma unroll
    for (int oy = 0; oy < TS; oy += 16) {
#pragma unroll
        for (int ox = 0; ox < TS; ox += 16) {
            const int cx = ox + tx, cy = oy + ty;
            int sx = 0, sy = 0;
#pragma unroll
            for (int k = 0; k < (RFIX ? 2 * RFIX : 1); k++) {
                if (RFIX) { sx += hx[(cy + k) * TS + cx]; sy += hy[(cy + k) * TS + cx]; }
            }
            if (!RFIX) for (int k = 0; k < 2 * r; k++) { sx += hx[(cy + k) * TS + cx]; sy += hy[(cy + k) * TS + cx]; }
            const int rx = (int)(int16_t)div_trunc(sx, d, log2d), ry = (int)(int16_t)div_trunc(sy, d, log2d);
            const int gx = blockIdx.x * TS + cx, gy = blockIdx.y * TS + cy;
            if (gx < lw && gy < lh) {
                const size_t q = (size_t)gy * lw + gx;
                blurred[q] = (int16_t)rx;
                blurred[(size_t)lw * lh + q] = (int16_t)ry;
                packed[q] = ((uint32_t)rx & 0xFFFFu) | ((uint32_t)ry << 16);
            }
        }
    }
}

__global__ void pack_flow_kernel(const int16_t* __restrict__ flow, uint32_t* __restrict__ packed, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) packed[i] = ((uint32_t)flow[i] & 0xFFFFu) | ((uint32_t)(uint16_t)flow[n + i] << 16);
}

// ------------------------------------------------------------------------------------------
// levels / warp / copy
// ------------------------------------------------------------------------------------------
struct Levels {
    float black, white, rcp_y, rcp_uv;
};
__device__ __forceinline__ Levels make_levels(float black, float white) {
    Levels l;
    l.black = black;
    l.white = white;
    l.rcp_y = __builtin_amdgcn_rcpf(white - black);
    l.rcp_uv = __builtin_amdgcn_rcpf(white);
    return l;
}
template <typename E>
__device__ __forceinline__ unsigned levels_y(float v, const Levels& l) {  // warpFrameKernelSDR.h:3-5
    using T = ElemTraits<E>;
    float f = ((v - l.black) * l.rcp_y) * T::maxv;
    f = fmaxf(fminf(f, T::maxv), 0.0f);
    return (unsigned)f & 0xFFFFu;
}
template <typename E>
__device__ __forceinline__ unsigned levels_uv(float v, const Levels& l) {  // warpFrameKernelSDR.h:7-9
    using T = ElemTraits<E>;
    float f = __builtin_fmaf((v - T::mid) * l.rcp_uv, T::maxv, T::mid);
    f = fmaxf(fminf(f, T::maxv), 0.0f);
    return (unsigned)f & 0xFFFFu;
}

__device__ __forceinline__ int mirror_warp(int pos, int dim) {  // warpFrameKernelSDR.h:12-20
    int res = pos;
    if (pos >= dim - 1) res = pos - ((pos - (dim - 2)) * 2);
    else if (pos < 1) res = -pos + 1;
    return min(max(res, 1), dim - 2);
}

// Diagnostic HSV view of the flow (output mode 3; semantics of warpFrameKernelSDR.h:23-113 / HDR :23-113): the direction of the
// offset picks a fully saturated hue, its magnitude the brightness, and the result is mixed half-and-half with the blended luma.
// A saturated hue wheel is one six-step ramp read at three rotations: at sextant k of the wheel a channel is
//     wheel(k) = 255, falling, 0, 0, rising, 255        (k = 0 .. 5)
// with red = wheel(k), green = wheel(k + 4), blue = wheel(k + 2) -- so the sextant is never branched on.
__device__ __forceinline__ unsigned hue_wheel(int k, unsigned rising, unsigned falling) {
    k = k >= 6 ? k - 6 : k;
    return (k == 0 || k == 5) ? 255u : k == 1 ? falling : k == 4 ? rising : 0u;
}
__device__ __forceinline__ unsigned to_byte(float v) { return (unsigned)fmaxf(fminf(v, 255.0f), 0.0f) & 0xFFu; }

template <typename E>
__device__ unsigned visualize_flow(int flow_x, int flow_y, unsigned blended, int channel, int gain) {
    using T = ElemTraits<E>;
    const int16_t vx = (int16_t)flow_x, vy = (int16_t)flow_y;
    const unsigned mag_x = (unsigned)(vx < 0 ? -vx : vx) & 0xFFFFu, mag_y = (unsigned)(vy < 0 ? -vy : vy) & 0xFFFFu;
    unsigned rgb[3] = {0u, 0u, 0u};
    if (mag_x != 0u || mag_y != 0u) {                         // (a zero offset stays black)
        float deg = atan2f((float)vy, (float)vx) * (180.0f / 3.14159274101257f);
        if (deg < 0) deg += 360.0f;
        deg = fmodf(deg, 360.0f);
        if (deg < 0) deg += 360.0f;
        const float wheel_pos = deg / 360.0f * 6.0f;          // 0 .. 6 around the hue wheel
        const int sextant = (int)wheel_pos;
        const float frac = wheel_pos - (float)sextant;
        const unsigned rising = (unsigned)(frac * 255.0f) & 0xFFu, falling = (unsigned)((1.0f - frac) * 255.0f) & 0xFFu;
        const int k = sextant % 6;
        // brightness: red and blue scale with |x| + |y|, green with twice |y| (the reference's weights)
        const float scale[3] = {(float)((int)mag_x + (int)mag_y), (float)mag_y * 2.0f, (float)((int)mag_x + (int)mag_y)};
        const int rot[3] = {0, 4, 2};
#pragma unroll
        for (int c = 0; c < 3; c++)
            rgb[c] = to_byte((float)hue_wheel(k + rot[c], rising, falling) / 255.0f * scale[c] * (float)gain);
    }
    const float fr = (float)rgb[0], fg = (float)rgb[1], fb = (float)rgb[2];
    if (channel == 0) {                                       // BT.601 luma of the colour, averaged with the blended picture
        const unsigned y = (unsigned)fmaxf(fminf(fr * 0.299f + fg * 0.587f + fb * 0.114f, 255.0f), 0.0f);
        if (T::hdr) return (((y & 0xFFFFu) << 7) + (blended >> 1)) & 0xFFFFu;
        return (((y & 0xFFu) >> 1) + ((blended & 0xFFu) >> 1)) & 0xFFu;
    }
    const float chroma = channel == 1 ? fr * -0.168736f + fg * -0.331264f + fb * 0.5f + 128.0f
                                      : fr * 0.5f + fg * -0.418688f + fb * -0.081312f + 128.0f;
    const unsigned cb = (unsigned)fmaxf(fminf(chroma, 255.0f), 0.0f);
    return T::hdr ? ((cb & 0xFFFFu) << 8) & 0xFFFFu : cb & 0xFFu;
}

struct WarpArgs {
    const void* frame12;
    const void* frame21;
    const int16_t* flow;   // blurred flow [2][lh][lw]
    const uint32_t* flow_xy; // same flow packed x | y << 16, [lh][lw]
    void* out;
    float s12, s21;        // frameScalar12 = t, frameScalar21 = 1 - t (opticalFlowCalcSDR.cpp:149-150)
    int mode;
    float black, white;
    // fast kernel: the outputs of one source period produced in ONE pass (flow looked up once, source rows
    // re-read from L1/L2 instead of HBM).  n_out == 1 for a plain warpFrames call.
    int n_out;
    float s12v[kMaxWarpOutputs], s21v[kMaxWarpOutputs];
    void* outv[kMaxWarpOutputs];
    uint32_t* plane21;     // warp_wg_kernel: also build the full phase plane of frame21 here (nullptr: no) -- see emit_plane_rows
};
// The fused period launches of up to kMaxFlowBatch contexts of one geometry as ONE launch (hf_batch): the unit index
// selects the member, so a batch of 8 pairs is one bandwidth-bound launch instead of 8 serialised ones.
struct WarpBatchArgs {
    int n;
    uint32_t* counters;     // diagnostic counters of the launch (hf_kernels.h kCounterWarp), nullptr: none
    WarpArgs s[kMaxWarpBatch];
};

// One output element of warpFrameKernel (all modes).
template <typename E>
__device__ __forceinline__ unsigned warp_element(const Geom& g, const WarpArgs& a, const Levels& lv, int cz, int cx, int cy) {
    using T = ElemTraits<E>;
    const E* __restrict__ A = (const E*)a.frame12;
    const E* __restrict__ B = (const E*)a.frame21;
    const int H = g.H, W = g.W, Si = g.in_stride, rs = g.rs, lw = g.lw, lh = g.lh;
    const size_t plane = (size_t)cz * H * Si;
    int ax = cx, ay = cy;
    const int mode = a.mode;
    HF_DBG_CHECK(cx >= 0 && cx < W && cy >= 0 && cy < (cz ? (H >> 1) : H), 205);
    if (mode == 5 && cx < (W >> 1)) return A[plane + (size_t)cy * Si + cx];  // :133-135
    if (mode == 6) {                                                          // :136-150
        const int vo = (H >> 2) >> cz;
        const bool in_rows = cy >= vo && cy < vo + (H >> (1 + cz));
        HF_DBG_CHECK(!(in_rows && cx < (W >> 1)) || (((cy - vo) << 1) < (cz ? (H >> 1) : H) && (cx << 1) + 1 < W), 206);
        if (in_rows && cx < (W >> 1)) return A[plane + (size_t)((cy - vo) << 1) * Si + (cx << 1) + (cz ? (cx & 1) : 0)];
        if (in_rows && cx < W) { ax = (cx - (W >> 1)) << 1; ay = (cy - vo) << 1; }
        else return cz ? T::midu : 0u;
    }
    const size_t N = (size_t)lw * lh;
    const int lx = cz ? ((ax >> rs) & ~1) : (ax >> rs);   // :153-154
    const int ly = cz ? ((ay >> rs) << 1) : (ay >> rs);
    HF_DBG_CHECK(lx >= 0 && ly >= 0 && lx < lw && ly < lh, 201);
    const int ox12 = a.flow[(size_t)ly * lw + lx];
    const int oy12 = a.flow[N + (size_t)ly * lw + lx];
    const int py = clampi(ly - (oy12 >> rs), 0, lh - 1);  // arithmetic shift, :157-158
    const int px = clampi(lx - (ox12 >> rs), 0, lw - 1);
    const int ox21 = a.flow[(size_t)py * lw + px];
    const int oy21 = a.flow[N + (size_t)py * lw + px];
    if (mode == 4) {                                      // :161-164
        if (cz) return T::midu;
        const unsigned mag = (unsigned)abs(ox12) + (unsigned)abs(oy12);
        return T::hdr ? min(mag << 10, 65535u) : min(mag << 2, 255u);
    }
    const float hy = cz ? 0.5f : 1.0f;
    const int dim_y = cz ? (H >> 1) : H;
    const int x12 = mirror_warp(ax + (int)roundf((float)ox12 * a.s12), W);           // :167-170
    const int y12 = mirror_warp(ay + (int)roundf((float)oy12 * a.s12 * hy), dim_y);
    const int x21 = mirror_warp(ax - (int)roundf((float)ox21 * a.s21), W);
    const int y21 = mirror_warp(ay - (int)roundf((float)oy21 * a.s21 * hy), dim_y);
    const int par = cz ? (cx & 1) : 0;
    if (mode == 0) return A[plane + (size_t)y12 * Si + (cz ? (x12 & ~1) : x12) + par];
    if (mode == 1) return B[plane + (size_t)y21 * Si + (cz ? (x21 & ~1) : x21) + par];
    HF_DBG_CHECK(x12 >= 0 && x21 >= 0 && y12 >= 0 && y21 >= 0 && y12 < dim_y && y21 < dim_y && (cz ? (x12 & ~1) + 1 : x12) < W && (cz ? (x21 & ~1) + 1 : x21) < W, 202);
    const float fa = (float)A[plane + (size_t)y12 * Si + (cz ? (x12 & ~1) : x12) + par];
    const float fb = (float)B[plane + (size_t)y21 * Si + (cz ? (x21 & ~1) : x21) + par];
    unsigned blended = (unsigned)__builtin_fmaf(fa, a.s21, fb * a.s12) & 0xFFFFu;     // :176-177 as compiled on gfx950
    if (mode == 3) blended = visualize_flow<E>(-ox12, -oy12, blended, cz + par, rs <= 2 ? 4 : 1);
    return cz ? levels_uv<E>((float)blended, lv) : levels_y<E>((float)blended, lv);
}

// Each thread produces VEC consecutive elements of one row and stores them with one wide store.
template <typename E, int VEC, bool ALIGNED>
__global__ __launch_bounds__(256) void warp_kernel(const Geom g, const WarpArgs a) {
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);  // 0 .. H + H/2 - 1 : Y rows then UV rows
    const int cx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * VEC;
    const int rows_total = g.H + (g.H >> 1);
    if (row >= rows_total || cx0 >= g.W) return;
    const int cz = row >= g.H;
    const int cy = cz ? row - g.H : row;
    const Levels lv = make_levels(a.black, a.white);
    E* __restrict__ out = (E*)a.out + (size_t)cz * g.H * g.out_stride + (size_t)cy * g.out_stride + cx0;
    if (ALIGNED && cx0 + VEC <= g.W) {
        static_assert(sizeof(E) * VEC == 16, "one 16-byte store per thread");
        __attribute__((aligned(16))) E v[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) v[i] = (E)warp_element<E>(g, a, lv, cz, cx0 + i, cy);
        store16_streaming(out, v);
    } else {
        for (int i = 0; i < VEC && cx0 + i < g.W; i++) out[i] = (E)warp_element<E>(g, a, lv, cz, cx0 + i, cy);
    }
}


// Fast path (modes 0-2, rs >= 1): the GROUP consecutive elements a thread handles at once share one
// low-res flow cell (GROUP divides 2^rs), so the flow is looked up ONCE per group (one packed load per
// direction) and, away from the left/right frame edges where mirrorCoordinate acts, the two source
// runs are contiguous: one unaligned GROUP-wide load each (two for a chroma run displaced by an odd
// amount).  Groups that touch the mirror zone fall back to warp_element.
template <typename E, int GROUP>
struct Run { __attribute__((aligned(sizeof(E) * GROUP))) E v[GROUP]; };

template <typename E, int GROUP>
__device__ __forceinline__ Run<E, GROUP> load_run(const E* __restrict__ p) {
    Run<E, GROUP> r;
    __builtin_memcpy(r.v, p, sizeof(E) * GROUP);   // unaligned global_load_dword{,x2,x4}
    return r;
}

// chroma run starting at x_first = cx + d: element i reads (x & ~1) + (i & 1) with x = x_first + i
// (warpFrameKernelSDR.h:173).  With e = x_first & ~1 and o = x_first & 1 the even slots are
// src[e + i] and the odd slots src[e + 2o + i]: one GROUP-wide run at e plus ONE extra pair (the two
// elements that follow the run when o = 1), no lane divergence.
template <typename E, int GROUP>
__device__ __forceinline__ Run<E, GROUP> load_run_uv(const E* __restrict__ row, int x_first) {
    const int e = x_first & ~1, o = x_first & 1;
    const Run<E, GROUP> lo = load_run<E, GROUP>(row + e);
    const Run<E, 2> ext = load_run<E, 2>(row + e + GROUP - 2 + 2 * o);   // o = 0: re-reads the run's tail (unused)
    Run<E, GROUP> r;
#pragma unroll
    for (int i = 0; i < GROUP; i++) {
        if ((i & 1) == 0) r.v[i] = lo.v[i];
        else r.v[i] = o ? (i + 2 < GROUP ? lo.v[(i + 2) % GROUP] : ext.v[(i + 2) % GROUP % 2]) : lo.v[i];
    }
    return r;
}

// The same runs through DWORD-ALIGNED loads + a funnel shift.  Measured on MI355X (tools/ubench/gather_rate.hip,
// L1-resident data, clocks per wave instruction): global_load_dwordx4 at a 4-byte aligned address 17.5, at a
// 2-byte aligned one 65.8; dword 6.1 vs 17.3; dwordx2 17.3 vs 33.4 -- every access that is not dword-aligned takes
// a slow path in the texture addresser.  A displaced run starts at an arbitrary element, so half (16-bit) or three
// quarters (8-bit) of the plain loads were slow ones.  Here: NDW + 1 aligned dwords, v_alignbit_b32 per dword.
// Requires a dword-aligned frame base and row pitch, and W * sizeof(E) % 4 == 0 (every dword that holds a needed
// byte then lies inside the row; the one extra dword that holds none when the run IS aligned is redirected).
template <typename E, int GROUP>
__device__ __forceinline__ Run<E, GROUP> load_run_dw(const unsigned char* __restrict__ row, int x) {
    constexpr int NDW = GROUP * (int)sizeof(E) / 4;
    const unsigned boff = (unsigned)x * (unsigned)sizeof(E);
    const unsigned sh = (boff & 3u) * 8u;
    const uint32_t* __restrict__ p = (const uint32_t*)(row + (boff & ~3u));
    struct __attribute__((aligned(4))) DW { uint32_t d[NDW]; } w;
    __builtin_memcpy(&w, p, sizeof(w));
    const uint32_t ext = p[sh ? NDW : 0];
    uint32_t o[NDW];
#pragma unroll
    for (int j = 0; j < NDW; j++) o[j] = __builtin_amdgcn_alignbit(j + 1 < NDW ? w.d[j + 1] : ext, w.d[j], sh);
    Run<E, GROUP> r;
    __builtin_memcpy(r.v, o, sizeof(o));
    return r;
}
// Chroma: elements [e, e + GROUP + 1] (the run at e plus the pair behind it) are always inside the row for an
// interior run; slot i takes element e + i (i even) or e + 2o + i (i odd): one v_perm_b32 per dword.
template <typename E, int GROUP>
__device__ __forceinline__ Run<E, GROUP> load_run_uv_dw(const unsigned char* __restrict__ row, int x_first) {
    constexpr int NDW = GROUP * (int)sizeof(E) / 4;
    const int e = x_first & ~1;
    const unsigned odd = (unsigned)x_first & 1u;
    const unsigned boff = (unsigned)e * (unsigned)sizeof(E);
    const unsigned sh = (boff & 3u) * 8u;          // 0 for 16-bit elements, 0 or 16 for 8-bit ones
    const uint32_t* __restrict__ p = (const uint32_t*)(row + (boff & ~3u));
    struct __attribute__((aligned(4))) DW { uint32_t d[NDW + 1]; } w;
    __builtin_memcpy(&w, p, sizeof(w));
    uint32_t L[NDW + 1];
#pragma unroll
    for (int j = 0; j < NDW; j++) L[j] = __builtin_amdgcn_alignbit(w.d[j + 1], w.d[j], sh);
    L[NDW] = w.d[NDW] >> sh;
    const uint32_t sel = sizeof(E) == 2 ? (odd ? 0x07060100u : 0x03020100u) : (odd ? 0x05020300u : 0x03020100u);
    uint32_t o[NDW];
#pragma unroll
    for (int j = 0; j < NDW; j++) o[j] = __builtin_amdgcn_perm(L[j + 1], L[j], sel);
    Run<E, GROUP> r;
    __builtin_memcpy(r.v, o, sizeof(o));
    return r;
}

// The same runs through BUFFER loads: resource descriptor (plane base, size) in SGPRs, a 32-bit byte offset per lane -- no
// 64-bit address arithmetic on the vector ALU (the flat-pointer form spent v_mad_i64_i32 + v_lshl_add_u64 pairs on every
// run), reads past the plane return 0 instead of faulting.  `off` = byte offset of the run's first element (luma) / of
// the even element e (chroma); dword-aligned plane base required.
typedef unsigned buf_v2 __attribute__((ext_vector_type(2)));
typedef unsigned buf_v4 __attribute__((ext_vector_type(4)));
template <int NDW>
__device__ __forceinline__ void buffer_load_dwords(uint32_t* d, __amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    static_assert(NDW == 1 || NDW == 2 || NDW == 4, "dword, dwordx2 or dwordx4");
    if constexpr (NDW == 4) {
        const buf_v4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else if constexpr (NDW == 2) {
        const buf_v2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
        d[0] = v.x; d[1] = v.y;
    } else {
        d[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
    }
}
// The run from its NDW + 1 dwords w[] (w[0] = the dword that holds the run's first byte): funnel shift by the byte offset inside
// the dword; chroma (CZ) additionally picks element e + i (i even) / e + 2 * odd + i (i odd): one v_perm_b32 per dword.
template <typename E, int G, int CZ>
__device__ __forceinline__ Run<E, G> run_from_dwords(const uint32_t* w, unsigned off, unsigned odd) {
    constexpr int NDW = G * (int)sizeof(E) / 4;
    const unsigned sh = (off & 3u) * 8u;
    uint32_t o[NDW];
    if constexpr (CZ == 0) {
#pragma unroll
        for (int j = 0; j < NDW; j++) o[j] = __builtin_amdgcn_alignbit(w[j + 1], w[j], sh);
    } else {
        uint32_t L[NDW + 1];
        if constexpr (sizeof(E) == 2) {   // a 16-bit chroma run starts at an even element: always dword-aligned (sh == 0)
#pragma unroll
            for (int j = 0; j <= NDW; j++) L[j] = w[j];
        } else {
#pragma unroll
            for (int j = 0; j < NDW; j++) L[j] = __builtin_amdgcn_alignbit(w[j + 1], w[j], sh);
            L[NDW] = w[NDW] >> sh;
        }
        const uint32_t sel = sizeof(E) == 2 ? (odd ? 0x07060100u : 0x03020100u) : (odd ? 0x05020300u : 0x03020100u);
#pragma unroll
        for (int j = 0; j < NDW; j++) o[j] = __builtin_amdgcn_perm(L[j + 1], L[j], sel);
    }
    Run<E, G> r;
    __builtin_memcpy(r.v, o, sizeof(o));
    return r;
}
template <typename E, int G, int CZ>
__device__ __forceinline__ Run<E, G> get_run_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned off, unsigned odd, [[maybe_unused]] unsigned plane_bytes) {
    constexpr int NDW = G * (int)sizeof(E) / 4;
    const unsigned base = off & ~3u;
    HF_DBG_CHECK((size_t)base + 4 * (NDW + 1) <= (size_t)plane_bytes + 4, 204);   // (the last dword of a run at the very end of the plane holds no needed byte and reads as 0)
    uint32_t w[NDW + 1];
    buffer_load_dwords<NDW>(w, rsrc, base);
    w[NDW] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, base + 4u * NDW, 0, 0);
    return run_from_dwords<E, G, CZ>(w, off, odd);
}

// mirrorCoordinate without branches (same values as mirror_warp): low side max(p, 1 - p), high side min(., 2d - 4 - p)
__device__ __forceinline__ int mirror_warp_bl(int pos, int dim) {
    const int r = min(max(pos, 1 - pos), 2 * dim - 4 - pos);
    return min(max(r, 1), dim - 2);
}

// GROUP elements of a plane row starting at element x (luma) / the chroma run for x_first = x
template <typename E, int G, int CZ, bool DW>
__device__ __forceinline__ Run<E, G> get_run(const E* __restrict__ rowp, int x, [[maybe_unused]] int W = 1 << 30) {
    HF_DBG_CHECK(x >= 0 && (CZ ? (x & ~1) : x) + G + (CZ ? 2 : 0) <= W + (CZ ? 2 : 0), 209);   // (a chroma run reads the pair behind it: still inside the row for interior runs)
    if constexpr (DW && G * sizeof(E) >= 4) {   // (2-byte groups -- 8-bit frames at rs = 1 -- keep the plain element loads)
        return CZ ? load_run_uv_dw<E, G>((const unsigned char*)rowp, x) : load_run_dw<E, G>((const unsigned char*)rowp, x);
    } else {
        return CZ ? load_run_uv<E, G>(rowp, x) : load_run<E, G>(rowp + x);
    }
}

// Body of the fast path for one plane (CZ = 0 luma, 1 chroma), everything plane-dependent is
// compile-time so the per-element code is straight-line.
typedef float float2v __attribute__((ext_vector_type(2)));

template <int VB> struct StoreVec;
template <> struct StoreVec<16> { using type = uint4; };
template <> struct StoreVec<8> { using type = uint2; };

// Blend + levels + store of one output: ROWS rows x VEC elements per thread from the source runs S (MODE 0 / 1: the run itself).
template <typename E, int GROUP, int ROWS, int NG>
struct WarpSrc { Run<E, GROUP> ra[ROWS][NG], rb[ROWS][NG]; };

// `store(r, v)`: writes the VB bytes v of row r (one wide streaming store).
template <typename E, int GROUP, int ROWS, int MODE, int CZ, int VB, typename Store>
__device__ __forceinline__ void warp_finish_to(const WarpSrc<E, GROUP, ROWS, (VB / (int)sizeof(E)) / GROUP>& S, const float s12t, const float s21t,
                                               const int nrows, const Levels& lv, const Store& store) {
    using T = ElemTraits<E>;
    constexpr int VEC = VB / sizeof(E);
    constexpr int NG = VEC / GROUP;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        if (r >= nrows) continue;
        __attribute__((aligned(VB))) E v[VEC];
#pragma unroll
        for (int k = 0; k < NG; k++) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < GROUP; i++) v[k * GROUP + i] = S.ra[r][k].v[i];
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < GROUP; i++) v[k * GROUP + i] = S.rb[r][k].v[i];
            } else {
                // Blend + levels two elements at a time on the packed fp32 pipe (v_pk_mul/fma/add_f32: the
                // same IEEE operations per element as the scalar form).  The launcher only selects this
                // kernel for 0 <= t <= 1 and non-degenerate levels, where
                //   (float)(unsigned short)x == trunc(x)   (0 <= x < 65536)   and
                //   fmax(fmin(f, max), 0)    == med3(f, 0, max)               (f is never NaN).
#pragma unroll
                for (int i = 0; i < GROUP; i += 2) {
                    const float2v fa = {(float)S.ra[r][k].v[i], (float)S.ra[r][k].v[i + 1]};
                    const float2v fb = {(float)S.rb[r][k].v[i], (float)S.rb[r][k].v[i + 1]};
                    const float2v s12 = {s12t, s12t}, s21 = {s21t, s21t};
                    float2v bl = __builtin_elementwise_fma(fa, s21, fb * s12);          // :176-177 as compiled on gfx950
                    bl.x = __builtin_truncf(bl.x);
                    bl.y = __builtin_truncf(bl.y);
                    float2v f;
                    if (CZ) {
                        const float2v mid = {T::mid, T::mid}, rcp = {lv.rcp_uv, lv.rcp_uv}, mx = {T::maxv, T::maxv};
                        f = __builtin_elementwise_fma((bl - mid) * rcp, mx, mid);       // warpFrameKernelSDR.h:7-9
                    } else {
                        const float2v bk = {lv.black, lv.black}, rcp = {lv.rcp_y, lv.rcp_y}, mx = {T::maxv, T::maxv};
                        f = ((bl - bk) * rcp) * mx;                                       // warpFrameKernelSDR.h:3-5
                    }
                    if (sizeof(E) == 2) {   // v_cvt_u32_f32 truncates and clamps negatives to 0, the pack saturates at 65535
                        typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
                        const ushort2v pk = __builtin_amdgcn_cvt_pk_u16((unsigned)f.x, (unsigned)f.y);
                        v[k * GROUP + i] = (E)pk.x;
                        v[k * GROUP + i + 1] = (E)pk.y;
                    } else {
                        v[k * GROUP + i] = (E)(unsigned)__builtin_amdgcn_fmed3f(f.x, 0.0f, T::maxv);
                        v[k * GROUP + i + 1] = (E)(unsigned)__builtin_amdgcn_fmed3f(f.y, 0.0f, T::maxv);
                    }
                }
            }
        }
        store(r, v);
    }
}
// ... through a flat pointer: output frames are not read back on the GPU, so the stores are streaming (non-temporal) ones.  (Non-temporal
// LOADS of frame N-2, whose last use this is, were much slower: 76 vs 52 us -- the outputs of a period re-read its rows through L2.)
template <typename E, int GROUP, int ROWS, int MODE, int CZ, int VB>
__device__ __forceinline__ void warp_finish(const WarpSrc<E, GROUP, ROWS, (VB / (int)sizeof(E)) / GROUP>& S, const float s12t, const float s21t,
                                            E* __restrict__ out, const int So, const int nrows, const Levels& lv) {
    typedef unsigned nt_vec __attribute__((ext_vector_type(VB / 4)));
    warp_finish_to<E, GROUP, ROWS, MODE, CZ, VB>(S, s12t, s21t, nrows, lv,
                                                 [&](const int r, const E* v) { __builtin_nontemporal_store(*(const nt_vec*)v, (nt_vec*)(out + (size_t)r * So)); });
}

template <typename E, int GROUP, int ROWS, int MODE, int CZ, int VB, bool DW>
__device__ __forceinline__ void warp_fast_body(const Geom& g, const WarpArgs& a, int cy0, int cx0, int ti0, int ti1) {
    using T = ElemTraits<E>;
    using SV = typename StoreVec<VB>::type;
    constexpr int VEC = VB / sizeof(E);
    constexpr int NG = VEC / GROUP;
    static_assert(VEC % GROUP == 0, "group must divide the per-thread vector");
    const int H = g.H, W = g.W, Si = g.in_stride, So = g.out_stride, rs = g.rs, lw = g.lw, lh = g.lh;
    const int dim_y = CZ ? (H >> 1) : H;
    const int nrows = min(ROWS, dim_y - cy0);
    const Levels lv = make_levels(a.black, a.white);
    if (cx0 + VEC > W) {  // ragged right edge
        for (int ti = ti0; ti < ti1; ti++) {
            WarpArgs at = a;
            at.s12 = a.s12v[ti]; at.s21 = a.s21v[ti]; at.out = a.outv[ti];
            E* __restrict__ o = (E*)at.out + (size_t)CZ * H * So + (size_t)cy0 * So + cx0;
            for (int r = 0; r < nrows; r++)
                for (int i = 0; i < VEC && cx0 + i < W; i++) o[(size_t)r * So + i] = (E)warp_element<E>(g, at, lv, CZ, cx0 + i, cy0 + r);
        }
        return;
    }
    const E* __restrict__ A = (const E*)a.frame12 + (size_t)CZ * H * Si;
    const E* __restrict__ B = (const E*)a.frame21 + (size_t)CZ * H * Si;
    const int ly = CZ ? ((cy0 >> rs) << 1) : (cy0 >> rs);    // same for all ROWS rows
    constexpr bool need_a = MODE != 1, need_b = MODE != 0;
    // interior waves fetch their runs through buffer loads (32-bit offsets into the plane)
    constexpr bool BUF = DW && GROUP * sizeof(E) >= 4;
    const unsigned pitch_b = (unsigned)Si * (unsigned)sizeof(E);
    const unsigned plane_bytes = (unsigned)dim_y * pitch_b;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)plane_bytes, 0x00020000);

    // flow of the thread's cells: looked up once, shared by every output of the period
    int ox12[NG], oy12[NG], ox21[NG], oy21[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const int cx = cx0 + k * GROUP;
        const int lx = CZ ? ((cx >> rs) & ~1) : (cx >> rs);
        HF_DBG_CHECK(lx >= 0 && ly >= 0 && lx < lw && ly < lh, 203);
        const uint32_t f12 = a.flow_xy[(size_t)ly * lw + lx];
        ox12[k] = (int)(int16_t)(f12 & 0xFFFFu); oy12[k] = (int)(int16_t)(f12 >> 16);
        const int py = clampi(ly - (oy12[k] >> rs), 0, lh - 1);
        const int px = clampi(lx - (ox12[k] >> rs), 0, lw - 1);
        const uint32_t f21 = a.flow_xy[(size_t)py * lw + px];
        ox21[k] = (int)(int16_t)(f21 & 0xFFFFu); oy21[k] = (int)(int16_t)(f21 >> 16);
    }

  // Source runs of one output.  The period loop is software-pipelined: the runs of output ti + 1 are requested
  // before output ti is blended and stored, so the loads' latency overlaps the fp32 work and the stores.
  using Src = WarpSrc<E, GROUP, ROWS, NG>;
  auto issue = [&](const int ti, Src& S) {
    const float s12t = a.s12v[ti], s21t = a.s21v[ti];
    int xa[NG], xb[NG], dya[NG], dyb[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const int cx = cx0 + k * GROUP;
        xa[k] = cx + (int)roundf((float)ox12[k] * s12t);
        xb[k] = cx - (int)roundf((float)ox21[k] * s21t);
        if (CZ) {
            dya[k] = (int)roundf((float)oy12[k] * s12t * 0.5f);
            dyb[k] = -(int)roundf((float)oy21[k] * s21t * 0.5f);
        } else {
            dya[k] = (int)roundf((float)oy12[k] * s12t);
            dyb[k] = -(int)roundf((float)oy21[k] * s21t);
        }
    }

    // mirrorCoordinate is the identity on [1, W-2]: a run that stays inside is contiguous.  The test is
    // made wave-uniform so that interior waves (all but the first/last of a row) carry no edge code.
    bool interior = true;
#pragma unroll
    for (int k = 0; k < NG; k++) {
        if (need_a) interior = interior && xa[k] >= 1 && xa[k] + GROUP - 1 <= W - 2;
        if (need_b) interior = interior && xb[k] >= 1 && xb[k] + GROUP - 1 <= W - 2;
    }
    // Several flow cells per thread (NG > 1, e.g. 8-bit frames at rs = 2): where the blurred flow is locally
    // uniform the NG runs of a row are one contiguous VEC-wide run -> one 16-byte load instead of NG small ones.
    bool merged = NG > 1;
#pragma unroll
    for (int k = 1; k < NG; k++) {
        if (need_a) merged = merged && xa[k] == xa[0] + k * GROUP && dya[k] == dya[0];
        if (need_b) merged = merged && xb[k] == xb[0] + k * GROUP && dyb[k] == dyb[0];
    }
    // rows: mirrorCoordinate is the identity on [1, dim_y - 2] -- one wave-uniform test per output instead of a
    // reflection per row and frame
    bool y_inside = true;
    const int cy_lo = cy0, cy_hi = min(cy0 + ROWS - 1, dim_y - 1);
#pragma unroll
    for (int k = 0; k < NG; k++) {
        if (need_a) y_inside = y_inside && cy_lo + dya[k] >= 1 && cy_hi + dya[k] <= dim_y - 2;
        if (need_b) y_inside = y_inside && cy_lo + dyb[k] >= 1 && cy_hi + dyb[k] <= dim_y - 2;
    }
    const bool all_y_inside = __builtin_amdgcn_ballot_w64(!y_inside) == 0;
    if constexpr (BUF) if (all_y_inside && __builtin_amdgcn_ballot_w64(!interior) == 0) {
        // the common case, wave-uniform: every run of every lane lies where mirrorCoordinate is the identity in x and y
        const bool one_run = NG > 1 && __builtin_amdgcn_ballot_w64(!merged) == 0;
        const unsigned row1 = (cy0 + 1 < dim_y) ? pitch_b : 0u;     // ROWS == 2; a row past the plane end re-reads the last row (not stored)
        static_assert(ROWS == 2, "row1");
        if (one_run) {
            const unsigned oa = __umul24((unsigned)(cy0 + dya[0]), pitch_b) + (unsigned)(CZ ? (xa[0] & ~1) : xa[0]) * (unsigned)sizeof(E);
            const unsigned ob = __umul24((unsigned)(cy0 + dyb[0]), pitch_b) + (unsigned)(CZ ? (xb[0] & ~1) : xb[0]) * (unsigned)sizeof(E);
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                if (need_a) {
                    const Run<E, VEC> w = get_run_buf<E, VEC, CZ>(rsrcA, oa + (r ? row1 : 0u), (unsigned)xa[0] & 1u, plane_bytes);
#pragma unroll
                    for (int k = 0; k < NG; k++)
#pragma unroll
                        for (int i = 0; i < GROUP; i++) S.ra[r][k].v[i] = w.v[k * GROUP + i];
                }
                if (need_b) {
                    const Run<E, VEC> w = get_run_buf<E, VEC, CZ>(rsrcB, ob + (r ? row1 : 0u), (unsigned)xb[0] & 1u, plane_bytes);
#pragma unroll
                    for (int k = 0; k < NG; k++)
#pragma unroll
                        for (int i = 0; i < GROUP; i++) S.rb[r][k].v[i] = w.v[k * GROUP + i];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NG; k++) {
                const unsigned oa = __umul24((unsigned)(cy0 + dya[k]), pitch_b) + (unsigned)(CZ ? (xa[k] & ~1) : xa[k]) * (unsigned)sizeof(E);
                const unsigned ob = __umul24((unsigned)(cy0 + dyb[k]), pitch_b) + (unsigned)(CZ ? (xb[k] & ~1) : xb[k]) * (unsigned)sizeof(E);
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    if (need_a) S.ra[r][k] = get_run_buf<E, GROUP, CZ>(rsrcA, oa + (r ? row1 : 0u), (unsigned)xa[k] & 1u, plane_bytes);
                    if (need_b) S.rb[r][k] = get_run_buf<E, GROUP, CZ>(rsrcB, ob + (r ? row1 : 0u), (unsigned)xb[k] & 1u, plane_bytes);
                }
            }
        }
        return;
    }
    auto row_of = [&](const int p) { return all_y_inside ? p : mirror_warp_bl(p, dim_y); };
    if (NG > 1 && __builtin_amdgcn_ballot_w64(!(interior && merged)) == 0) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int cy = min(cy0 + r, dim_y - 1);
            if (need_a) {
                const E* rowp = A + (size_t)row_of(cy + dya[0]) * Si;
                const Run<E, VEC> w = get_run<E, VEC, CZ, DW>(rowp, xa[0], W);
#pragma unroll
                for (int k = 0; k < NG; k++)
#pragma unroll
                    for (int i = 0; i < GROUP; i++) S.ra[r][k].v[i] = w.v[k * GROUP + i];
            }
            if (need_b) {
                const E* rowp = B + (size_t)row_of(cy + dyb[0]) * Si;
                const Run<E, VEC> w = get_run<E, VEC, CZ, DW>(rowp, xb[0], W);
#pragma unroll
                for (int k = 0; k < NG; k++)
#pragma unroll
                    for (int i = 0; i < GROUP; i++) S.rb[r][k].v[i] = w.v[k * GROUP + i];
            }
        }
    } else if (__builtin_amdgcn_ballot_w64(!interior) == 0) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int cy = min(cy0 + r, dim_y - 1);           // rows past the plane end re-read the last row (not stored)
#pragma unroll
            for (int k = 0; k < NG; k++) {
                if (need_a) {
                    const E* rowp = A + (size_t)row_of(cy + dya[k]) * Si;
                    S.ra[r][k] = get_run<E, GROUP, CZ, DW>(rowp, xa[k], W);
                }
                if (need_b) {
                    const E* rowp = B + (size_t)row_of(cy + dyb[k]) * Si;
                    S.rb[r][k] = get_run<E, GROUP, CZ, DW>(rowp, xb[k], W);
                }
            }
        }
    } else {   // some lane of this wave touches the mirror zone at the left/right frame edge:
               // those lanes gather per element, the others keep their contiguous runs
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int cy = min(cy0 + r, dim_y - 1);
#pragma unroll
            for (int k = 0; k < NG; k++) {
                if (need_a) {
                    const E* rowp = A + (size_t)row_of(cy + dya[k]) * Si;
                    if (xa[k] >= 1 && xa[k] + GROUP - 1 <= W - 2) {
                        S.ra[r][k] = get_run<E, GROUP, CZ, DW>(rowp, xa[k], W);
                    } else {
#pragma unroll
                        for (int i = 0; i < GROUP; i++) {
                            const int x = mirror_warp(xa[k] + i, W);
                            HF_DBG_CHECK(x >= 0 && (CZ ? (x & ~1) + 1 : x) < W && rowp >= A && rowp < A + (size_t)dim_y * Si, 207);
                            S.ra[r][k].v[i] = rowp[CZ ? (x & ~1) + (i & 1) : x];
                        }
                    }
                }
                if (need_b) {
                    const E* rowp = B + (size_t)row_of(cy + dyb[k]) * Si;
                    if (xb[k] >= 1 && xb[k] + GROUP - 1 <= W - 2) {
                        S.rb[r][k] = get_run<E, GROUP, CZ, DW>(rowp, xb[k], W);
                    } else {
#pragma unroll
                        for (int i = 0; i < GROUP; i++) {
                            const int x = mirror_warp(xb[k] + i, W);
                            HF_DBG_CHECK(x >= 0 && (CZ ? (x & ~1) + 1 : x) < W && rowp >= B && rowp < B + (size_t)dim_y * Si, 208);
                            S.rb[r][k].v[i] = rowp[CZ ? (x & ~1) + (i & 1) : x];
                        }
                    }
                }
            }
        }
    }
  };
  auto finish = [&](const int ti, const Src& S) {
    E* __restrict__ out = (E*)a.outv[ti] + (size_t)CZ * H * So + (size_t)cy0 * So + cx0;
    warp_finish<E, GROUP, ROWS, MODE, CZ, VB>(S, a.s12v[ti], a.s21v[ti], out, So, nrows, lv);
  };
  // (software pipelining -- requesting the runs of output ti + 1, of two outputs ahead, or of all outputs before blending
  //  output ti -- measured no faster since the dword-aligned loads: 49.9 / 52.8 / 53.8 us against 48.7 us for the plain
  //  loop with its higher occupancy)
  for (int ti = ti0; ti < ti1; ti++) {
    Src cur;
    issue(ti, cur);
    finish(ti, cur);
  }
}

// Thread = VEC consecutive elements x ROWS consecutive rows that share one flow-cell row (ROWS
// divides 2^rs).  Flow lookups and displacement maths are done once per GROUP and reused for the
// ROWS rows; all 2 * ROWS source runs are requested before any is consumed.
// Waves (= consecutive wave tiles of a tile row) per workgroup, chosen per launch: 4, or 16 for the batched periods of large
// frames.  Measured on MI355X, 2160p HDR pipeline (32 pair streams = 2 batches of 16), k frames/s: 67.0 / 67.6 / 68.9 / 68.7 /
// 67.8 / 68.8 / 70.0 with 4 / 5 / 6 / 8 / 10 / 15 / 16 waves (+2.6 % with one batch stream); a single period is 46 us with 4
// and 52 us with 16 (760 workgroups for 256 CUs), and the short one-output waves of frames up to 1080p lose 8 % with 16.
constexpr int kWarpWavesSmall = 4, kWarpWavesLarge = 16;
// (only the one-flow-cell-per-thread instances are compiled for 1,024-thread workgroups = at most 128 VGPRs: the others need more)
constexpr int warp_max_waves(size_t elem, int group, int vb) { return vb == 16 && group * (int)elem == 16 ? kWarpWavesLarge : kWarpWavesSmall; }
// VB = bytes of output per thread and row: 16, or 8 for small frames (<= 1080p 8-bit), where 16-byte threads
// leave too few waves to hide the per-wave latency chain (one round of fat waves: 9.4 us for 9.3 MB).
// Wave tile = kWarpTX lanes x kWarpTY row groups: (16 x VEC) elements wide, (4 x ROWS) rows high -- at 2160p HDR 128 pixels
// x 8 rows = 16 flow cells of one cell row.  (Round 1 used 64 lanes along x: 7.5 tiles per 3840-pixel row, of which the
// first and the last contain lanes whose runs reach into the mirror zone of warpFrameKernelSDR.h:12-20, so 27 % of all
// waves executed the per-element edge path next to the run path -- that, not the interior code, was most of the
// 1,940 VALU instructions per wave.  With 128-pixel tiles 2 of 30 tiles per row are edge tiles.)
constexpr int kWarpTX = 16, kWarpTY = 4;
constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }
// Deferred phase planes (see warp_wg_kernel): plane-building workgroups carried by a period warp launch.
struct PlaneOut {
    PhaseLayout pl;
    int blocks;                          // plane-building workgroups per super row (0: the launch builds no planes)
    FastDiv per_member, per_sr, wpr;     // unit index -> (member, super row, block): scalar divisions (hf_kernels.h)
    uint32_t* counters;                  // diagnostic counters (hf_kernels.h kCounterWarp), nullptr: none
};
template <typename E, int GROUP, int ROWS, int MODE, int VB, bool DW>
__global__ __launch_bounds__(64 * warp_max_waves(sizeof(E), GROUP, VB)) void warp_fast_kernel(const Geom g, const WarpBatchArgs batch, int y_groups, int out_chunk, int n_chunks) {
    constexpr int VEC = VB / sizeof(E);
    // Work decomposition: tiles are numbered row-major (luma tile rows first, then chroma: the plane test is a scalar
    // branch), a workgroup takes 4 or 16 consecutive tiles, and workgroups are dealt to the XCDs in contiguous bands: linear
    // block id b runs on XCD b % 8 (MI355X_MICROARCH.md "Workgroup dispatch"), so block b works on band (b % 8).
    // Measured on the 2160p HDR blend: 20.7 us with the naive 2-D grid (every XCD walks a column stripe) -> 19.0 us
    // banded.  Placement only affects speed.  Units are ordered (member, tile block, chunk): with a batch every XCD works
    // on whole members, i.e. each source frame is pulled into ONE L2.
    const int uv_groups = ((g.H >> 1) + ROWS - 1) / ROWS;
    const int wpr = (g.W + kWarpTX * VEC - 1) / (kWarpTX * VEC);            // wave tiles per tile row
    const int y_tiles = (y_groups + kWarpTY - 1) / kWarpTY, uv_tiles = (uv_groups + kWarpTY - 1) / kWarpTY;
    const int n_tiles = wpr * (y_tiles + uv_tiles);
    const int wpb = (int)(blockDim.x >> 6);                                 // waves per workgroup (launch_warp_fast)
    const int n_blocks = (n_tiles + wpb - 1) / wpb;                         // per member
    // out_chunk = outputs of the period one thread produces.  Large frames: all of them (the sources are read from HBM once);
    // small frames: fewer, so that a period is n_chunks times as many, shorter waves -- their sources come from L2 anyway and
    // a 1080p period has only ~3,000 wave tiles for 1,024 SIMDs.  The chunks of a tile run next to each other on one XCD.
    const int total = n_blocks * n_chunks * batch.n;
    const int per_band = (total + 7) >> 3;
    const int u = (blockIdx.x & 7) * per_band + (blockIdx.x >> 3);
    if (u >= total) return;
    const int chunk = u % n_chunks, v = u / n_chunks;
    const int member = v / n_blocks, blk = v - member * n_blocks;
    const WarpArgs& a = batch.s[member];
    const int ti0 = chunk * out_chunk, ti1 = min(a.n_out, ti0 + out_chunk);
    if (ti0 >= ti1) return;
    const int tile = blk * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int trow = tile / wpr, tcol = tile - trow * wpr;
    const int lane = threadIdx.x & 63;
    const int cx0 = (tcol * kWarpTX + (lane & (kWarpTX - 1))) * VEC;
    if (cx0 >= g.W) return;
    if (trow >= y_tiles) {
        const int rg = (trow - y_tiles) * kWarpTY + (lane / kWarpTX);
        if (rg < uv_groups) warp_fast_body<E, GROUP, ROWS, MODE, 1, VB, DW>(g, a, rg * ROWS, cx0, ti0, ti1);
    } else {
        const int rg = trow * kWarpTY + (lane / kWarpTX);
        if (rg < y_groups) warp_fast_body<E, GROUP, ROWS, MODE, 0, VB, DW>(g, a, rg * ROWS, cx0, ti0, ti1);
    }
}

// ------------------------------------------------------------------------------------------
// LDS-staged period warp, one window per WORKGROUP
// ------------------------------------------------------------------------------------------
// The outputs of a source period read almost the same source rows -- the displacement only grows with the blending scalar.
// Here the NW waves of a workgroup take NW vertically stacked 128 x 8 wave tiles (one tile column, 8 NW rows of one plane) and
// copy, ONCE per period, the window of each source frame that all their outputs read into LDS with direct-to-LDS buffer loads
// (16 bytes per lane, 1 KB per instruction, no VGPR round trip); every output then takes its runs from LDS (five dword reads
// per run + the same funnel shift / byte permute as the global path).  One window per workgroup instead of one per wave: the
// windows of vertically neighbouring tiles overlap by the vertical displacement range, so the copy fetches 1.15-1.2 x the tile
// (per wave: 1.55 x).  The window is found per workgroup (per-wave min / max over lanes and outputs of the runs' 16-byte chunks
// and rows -- packed 16-bit DPP butterfly -- combined through LDS); workgroups whose runs do not fit the LDS budget (fast or
// diverging motion), touch the mirror zone or contain a partial wave take the global path (warp_fast_body): same results.
// Shape of the staged kernel (all measured on the 2160p HDR pipeline, 2 batch streams of 16, k frames/s -- DESIGN.md appendix C):
//   waves (= vertically stacked wave tiles) per workgroup: 68.8-70.0 without staging, 71.9-72.5 / 72.9-73.4 / 73.9-74.1 / 66.7-67.8 / 61.3
//   with 2 / 3 / 4 / 6 / 16 waves;  rows per thread: 2 (4 rows = half the waves: alone 672 vs 663 us per 16 members, pipeline 69.4 vs
//   72.4: the other stream's chain waits longer for the fewer, longer waves);  window budget 160 / 176 / 192 / 224 / 256 chunks per wave:
//   192 best (smaller: more fallbacks; larger: one workgroup per CU less)
constexpr int kWgWaves = 4, kWgRows = 2, kWgChunksPerWave = 192;
constexpr long kWgMinWaves = 4 * 8192;            // the staged kernel only for launches of several rounds of waves (batched periods): ONE
                                                  // member's period alone is 12 % slower that way (two barriers and a serial prologue per
                                                  // workgroup with nothing to overlap them), so single launches keep the global path
constexpr int wg_chunks(int nw) { return nw * kWgChunksPerWave; }   // 16-byte chunks per source window (12 KB for 4 waves)

typedef unsigned short ushort2w __attribute__((ext_vector_type(2)));
// min / max of both unsigned 16-bit halves (v_pk_min_u16 / v_pk_max_u16)
template <bool MAX>
__device__ __forceinline__ uint32_t pk_mm_u16(uint32_t a, uint32_t b) {
    ushort2w x, y;
    __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    const ushort2w m = MAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y);
    uint32_t r;
    __builtin_memcpy(&r, &m, 4);
    return r;
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
// ... over the wave (all 64 lanes active): xor butterfly inside every 16-lane row on the DPP path (quad_perm 1032 / 2301,
// row_half_mirror, row_mirror), then the four rows on the scalar unit
template <bool MAX>
__device__ __forceinline__ uint32_t wave_pk_mm_u16(uint32_t v) {
    v = pk_mm_u16<MAX>(v, dpp_mov<0xB1>(v));
    v = pk_mm_u16<MAX>(v, dpp_mov<0x4E>(v));
    v = pk_mm_u16<MAX>(v, dpp_mov<0x141>(v));
    v = pk_mm_u16<MAX>(v, dpp_mov<0x140>(v));
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return pk_mm_u16<MAX>(pk_mm_u16<MAX>(r0, r1), pk_mm_u16<MAX>(r2, r3));
}

// Extended coordinates of a staged workgroup.  A run may start up to kExtX elements left of / kExtY rows above the plane (and end as far
// beyond it): mirrorCoordinate (warpFrameKernelSDR.h:12-20) folds such positions back into the frame, and the window of a workgroup at
// the frame edge holds the plane EXTENDED that way (ext[y][x] = plane[mirror(y)][mirror(x)]), so that edge tiles read contiguous runs
// from LDS like every other tile.  All packed (row, byte offset) words carry these biases; kExtX * sizeof(E) is a multiple of 16, so
// 16-byte chunks of the extended row are 16-byte chunks of the plane row.
constexpr int kExtX = 64, kExtY = 64;
constexpr int kWgCells = 64;                       // flow cells of one workgroup tile at most (2160p HDR luma: 16 x 4)

template <int NW>
struct WgShared {                                  // static LDS of warp_wg_kernel
    uint2 tab[kMaxWarpOutputs][kWgCells];          // per output and flow cell of the tile: displacement words of source A (x) and B (y)
    uint32_t bounds[NW][4];                        // per wave: packed min / max of its items' run starts, A then B
    int state[NW];                                 // per wave: 2 = no tile (past the plane's end), 1 = full tile, 0 = partial tile
    int item[NW];                                  // per wave: bit 0 = all its items stageable, bit 1 = all of them interior (no mirroring)
};

template <typename E, int MODE, int CZ, int NW, int ROWS>
__device__ __forceinline__ void warp_wg_body(const Geom& g, const WarpArgs& a, const int tx0, const int ty0, const bool lane_valid, const int wave,
                                             unsigned char* const lds, WgShared<NW>& sh, uint32_t* const counters) {
    constexpr int VEC = 16 / (int)sizeof(E), NDW = 4, CHUNKS = wg_chunks(NW * ROWS / 2), SZ = (int)sizeof(E);
    constexpr int TW = kWarpTX * VEC, TH = NW * kWarpTY * ROWS, NT = 64 * NW;      // tile of the workgroup (elements x rows), its threads
    constexpr bool need_a = MODE != 1, need_b = MODE != 0;
    const int H = g.H, W = g.W, Si = g.in_stride, So = g.out_stride, rs = g.rs, lw = g.lw, lh = g.lh;
    const int dim_y = CZ ? (H >> 1) : H;
    const int n = a.n_out;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int cx0 = tx0 + (int)(lane & (kWarpTX - 1)) * VEC, cy0 = ty0 + (wave * kWarpTY + (int)(lane / kWarpTX)) * ROWS;
    const uint64_t valid_mask = __builtin_amdgcn_ballot_w64(lane_valid);
    // full: every lane owns VEC whole elements of its row.  Phase C stores 16 bytes per lane unconditionally, so a lane that hangs over the
    // row's end (W % VEC != 0: its store would reach into the stride padding, which warpFrameKernelSDR.h:116-120 never writes) makes its
    // wave a PARTIAL one: the workgroup then takes the generic body, whose ragged lanes store element by element up to W.
    const bool present = valid_mask != 0, full = __builtin_amdgcn_ballot_w64(lane_valid && cx0 + VEC <= W) == ~0ull;          // wave-uniform

    // ---- phase A, per FLOW CELL instead of per lane.  All lanes of a flow cell share their displacements (a cell is 2^rs luma rows
    // high and one (luma) or two (chroma: lx & ~1) cells wide; a 16-byte thread lies inside one cell), and a run is
    //     (row, first byte) = (cy0 + dy, (cx0 + dx) * SZ),
    // so the workgroup computes ONE displacement word dy << 16 + dx * SZ (+ the parity of dx for chroma) per cell, output and source --
    // thread t takes cell t % cells and the outputs t / cells, t / cells + threads / cells, ... -- and leaves it in LDS; a lane adds its own
    // (cy0, cx0) word.  (Round 3 computed all 2 x n runs in every lane: four row-group lanes of a wave repeated the same arithmetic.)
    const int lcw = rs + CZ;                                                    // log2 of the cell width in elements
    const int lgx = max(0, ilog2c(TW) - lcw), lgy = max(0, ilog2c(TH) - rs);    // log2 of the cells per tile row / column
    const int lg = lgx + lgy;
    const int cw = min(1 << lcw, TW), ch = min(1 << rs, TH);                    // extent of a cell inside the tile
    uint32_t lo_a = 0xFFFFFFFFu, hi_a = 0u, lo_b = 0xFFFFFFFFu, hi_b = 0u;      // packed row << 16 | byte extremes of the run STARTS (biased)
    bool it_ok = lg <= ilog2c(kWgCells), it_in = true;
    {
        const int cell = (int)tid & ((1 << lg) - 1);
        const int cell_x0 = tx0 + ((cell & ((1 << lgx) - 1)) << lcw), cell_y0 = ty0 + ((cell >> lgx) << rs);
        if (it_ok && cell_x0 < W && cell_y0 < dim_y) {                          // (cells past the plane's end belong to waves without a tile)
            const int ly = min(CZ ? ((cell_y0 >> rs) << 1) : (cell_y0 >> rs), lh - 1);
            const int lx = min(CZ ? ((cell_x0 >> rs) & ~1) : (cell_x0 >> rs), lw - 1);
            HF_DBG_CHECK(lx >= 0 && ly >= 0 && cell < kWgCells, 210);
            const uint32_t f12 = a.flow_xy[(size_t)ly * lw + lx];
            const int ox12 = (int)(int16_t)(f12 & 0xFFFFu), oy12 = (int)(int16_t)(f12 >> 16);
            const int py = clampi(ly - (oy12 >> rs), 0, lh - 1), px = clampi(lx - (ox12 >> rs), 0, lw - 1);
            const uint32_t f21 = a.flow_xy[(size_t)py * lw + px];
            const int ox21 = (int)(int16_t)(f21 & 0xFFFFu), oy21 = (int)(int16_t)(f21 >> 16);
            // one source of one output: displacement word; extremes of the run starts over the lanes of the cell; stageable? interior?
            auto item = [&](const int dx, const int dy, uint32_t& lo, uint32_t& hi) -> uint32_t {
                const int dxe = CZ ? (dx & ~1) : dx;
                const int x_lo = cell_x0 + dxe, x_hi = x_lo + cw - VEC, y_lo = cell_y0 + dy, y_hi = y_lo + ch - ROWS;
                const int bx_lo = (x_lo + kExtX) * SZ, bx_hi = (x_hi + kExtX) * SZ, by_lo = y_lo + kExtY, by_hi = y_hi + kExtY;
                // inside the extended plane (every field of every packed word stays in 16 bits, run tails included); a chroma run that
                // reaches the RIGHT mirror zone is not stageable: there the element a slot reads depends on the parity of dx, i.e. the
                // extended chroma row is not a function of the position alone (on the left and for luma it is)
                bool ok = bx_lo >= 0 && bx_hi + 4 * NDW + 4 <= 0xFFFF && by_lo >= 0 && by_hi + ROWS <= 0xFFFF;
                if (CZ) ok = ok && x_hi + VEC <= W - 2;
                it_ok = it_ok && ok;
                it_in = it_in && x_lo >= 1 && x_hi + VEC - 1 + CZ <= W - 2 && y_lo >= 1 && y_hi + ROWS - 1 <= dim_y - 2;
                if (ok) {
                    lo = pk_mm_u16<false>(lo, ((uint32_t)by_lo << 16) | (uint32_t)bx_lo);
                    hi = pk_mm_u16<true>(hi, ((uint32_t)by_hi << 16) | (uint32_t)bx_hi);
                }
                return (uint32_t)dy * 65536u + (uint32_t)(dxe * SZ + (CZ ? (dx & 1) : 0));   // (unsigned: extreme offsets wrap instead of overflowing; they are not stageable anyway)
            };
            for (int j = (int)tid >> lg; j < n; j += NT >> lg) {
                float s12t = a.s12v[0], s21t = a.s21v[0];                      // (j differs between the lanes of a wave when the tile has < 64 cells)
#pragma unroll
                for (int k = 1; k < kMaxWarpOutputs; k++) { s12t = j == k ? a.s12v[k] : s12t; s21t = j == k ? a.s21v[k] : s21t; }
                uint2 w = make_uint2(0u, 0u);
                if (need_a) w.x = item((int)roundf((float)ox12 * s12t), CZ ? (int)roundf((float)oy12 * s12t * 0.5f) : (int)roundf((float)oy12 * s12t), lo_a, hi_a);
                if (need_b) w.y = item(-(int)roundf((float)ox21 * s21t), -(CZ ? (int)roundf((float)oy21 * s21t * 0.5f) : (int)roundf((float)oy21 * s21t)), lo_b, hi_b);
                sh.tab[j][cell] = w;
            }
        }
    }
    {   // per wave: extremes of its items, and what kind of tile it has
        const bool ok_all = __builtin_amdgcn_ballot_w64(!it_ok) == 0, in_all = __builtin_amdgcn_ballot_w64(!it_in) == 0;
        uint32_t b0 = 0xFFFFFFFFu, b1 = 0u, b2 = 0xFFFFFFFFu, b3 = 0u;
        if (need_a) { b0 = wave_pk_mm_u16<false>(lo_a); b1 = wave_pk_mm_u16<true>(hi_a); }
        if (need_b) { b2 = wave_pk_mm_u16<false>(lo_b); b3 = wave_pk_mm_u16<true>(hi_b); }
        // (selects, not an array indexed by the lane: that would live in scratch memory -- 16 bytes per lane of HBM traffic)
        const uint32_t mine = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
        if (lane < 4) sh.bounds[wave][lane] = mine;
        if (lane == 0) { sh.state[wave] = !present ? 2 : full ? 1 : 0; sh.item[wave] = (ok_all ? 1 : 0) | (in_all ? 2 : 0); }
    }
    __syncthreads();
    bool wg_ok = true, wg_in = true;
    uint32_t m0 = 0xFFFFFFFFu, m1 = 0u, m2 = 0xFFFFFFFFu, m3 = 0u;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        wg_ok = wg_ok && sh.state[w] != 0 && (sh.item[w] & 1) != 0;
        wg_in = wg_in && (sh.item[w] & 2) != 0;
        m0 = pk_mm_u16<false>(m0, sh.bounds[w][0]); m1 = pk_mm_u16<true>(m1, sh.bounds[w][1]);
        m2 = pk_mm_u16<false>(m2, sh.bounds[w][2]); m3 = pk_mm_u16<true>(m3, sh.bounds[w][3]);
    }
    const bool runs_ok = __builtin_amdgcn_readfirstlane((int)wg_ok) != 0;       // every wave full or absent, every run inside the extended plane
    wg_in = __builtin_amdgcn_readfirstlane((int)wg_in) != 0;
    // window of a source: 16-byte chunks [cmin, cmin + C) x rows [ymin, ymin + R) of the extended plane; a run spans NDW + 1 dwords from
    // the dword of its first byte
    int cmin_a = 0, ymin_a = 0, C_a = 1, R_a = 0, cmin_b = 0, ymin_b = 0, C_b = 1, R_b = 0;
    if (runs_ok) {
        const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)m0), ha = (uint32_t)__builtin_amdgcn_readfirstlane((int)m1);
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)m2), hb = (uint32_t)__builtin_amdgcn_readfirstlane((int)m3);
        if (need_a) {
            cmin_a = (int)(la & 0xFFFFu) >> 4; ymin_a = (int)(la >> 16);
            C_a = (int)(((ha & 0xFFFCu) + 4u * NDW + 3u) >> 4) - cmin_a + 1; R_a = (int)(ha >> 16) + ROWS - 1 - ymin_a + 1;
        }
        if (need_b) {
            cmin_b = (int)(lb & 0xFFFFu) >> 4; ymin_b = (int)(lb >> 16);
            C_b = (int)(((hb & 0xFFFCu) + 4u * NDW + 3u) >> 4) - cmin_b + 1; R_b = (int)(hb >> 16) + ROWS - 1 - ymin_b + 1;
        }
        // (a direct-to-LDS instruction writes 64 chunks: the windows are rounded up to that)
        wg_ok = ((R_a * C_a + 63) & ~63) <= CHUNKS && ((R_b * C_b + 63) & ~63) <= CHUNKS && C_a <= 64 && C_b <= 64;
    }
    wg_ok = __builtin_amdgcn_readfirstlane((int)wg_ok) != 0;
    // this lane's own word and cell: run of output j in source A = base + tab[j][ci].x (row << 16 | byte offset, biased; chroma: bit 0 = dx odd)
    const uint32_t base = ((uint32_t)(cy0 + kExtY) << 16) | (uint32_t)((cx0 + kExtX) * SZ);
    const int ci = (((cy0 - ty0) >> rs) << lgx) | ((cx0 - tx0) >> lcw);
    HF_DBG_CHECK(!wg_ok || (ci >= 0 && ci < kWgCells), 211);
    if (counters && threadIdx.x == 0) atomicAdd(counters + kCounterWarp + (wg_ok ? 0 : (runs_ok && wg_in && full) ? 1 : 2), 1u);
    if (!wg_ok) {   // workgroup-uniform: no barrier follows
        if (runs_ok && wg_in && full) {
            // every run of the workgroup is interior, only its window does not fit (fast or diverging motion): the global path straight
            // from the run table -- no second displacement pass, no edge tests
            const unsigned pitch_g = (unsigned)Si * (unsigned)SZ;
            const unsigned plane_g = (unsigned)dim_y * pitch_g;
            const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)((const E*)a.frame12 + (size_t)CZ * H * Si), 0, (int)plane_g, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)((const E*)a.frame21 + (size_t)CZ * H * Si), 0, (int)plane_g, 0x00020000);
            const Levels lvg = make_levels(a.black, a.white);
            const size_t out_g = (size_t)CZ * H * So + (size_t)cy0 * So + cx0;
            for (int j = 0; j < n; j++) {
                WarpSrc<E, VEC, ROWS, 1> S;
                const uint2 d = sh.tab[j][ci];
                if (need_a) {
                    const uint32_t w = base + d.x, wb = w & 0xFFFFu, odd = CZ ? (wb & 1u) : 0u;
                    const unsigned o = __umul24((w >> 16) - kExtY, pitch_g) + (CZ ? (wb & ~1u) : wb) - (unsigned)(kExtX * SZ);
                    HF_DBG_CHECK((o & ~3u) + 4u * (NDW + 1) + (ROWS - 1) * pitch_g <= plane_g, 10);
#pragma unroll
                    for (int r = 0; r < ROWS; r++) S.ra[r][0] = get_run_buf<E, VEC, CZ>(rsrcA, o + (unsigned)r * pitch_g, odd, plane_g);
                }
                if (need_b) {
                    const uint32_t w = base + d.y, wb = w & 0xFFFFu, odd = CZ ? (wb & 1u) : 0u;
                    const unsigned o = __umul24((w >> 16) - kExtY, pitch_g) + (CZ ? (wb & ~1u) : wb) - (unsigned)(kExtX * SZ);
                    HF_DBG_CHECK((o & ~3u) + 4u * (NDW + 1) + (ROWS - 1) * pitch_g <= plane_g, 11);
#pragma unroll
                    for (int r = 0; r < ROWS; r++) S.rb[r][0] = get_run_buf<E, VEC, CZ>(rsrcB, o + (unsigned)r * pitch_g, odd, plane_g);
                }
                warp_finish<E, VEC, ROWS, MODE, CZ, 16>(S, a.s12v[j], a.s21v[j], (E*)a.outv[j] + out_g, So, ROWS, lvg);
            }
        } else if (lane_valid) {   // partial tiles, runs beyond the extended plane, chroma in the right mirror zone: the generic body
#pragma unroll
            for (int r = 0; r < ROWS; r += 2)   // (the generic body takes two rows)
                if (cy0 + r < dim_y) warp_fast_body<E, VEC, 2, MODE, CZ, 16, true>(g, a, cy0 + r, cx0, 0, n);
        }
        return;
    }

    // ---- phase B: copy the windows.  Chunk q of a window lies at row q / C, column q % C; its LDS address is 16 q.
    const unsigned pitch_b = (unsigned)Si * (unsigned)SZ;
    const unsigned plane_bytes = (unsigned)dim_y * pitch_b;
    const E* __restrict__ A = (const E*)a.frame12 + (size_t)CZ * H * Si;
    const E* __restrict__ B = (const E*)a.frame21 + (size_t)CZ * H * Si;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // AUX: cache policy of the copy.  Frame N-2 (source A) is read for the last time here: its window copy is non-temporal (aux 2) so that its
    // lines leave L2 / the Infinity Cache first -- +0.9-1.2 % frames/s on the 2160p HDR pipeline, alternating on one box; the same hint on
    // source B, which the plane-building workgroups of this launch read too, costs 0.7-1.6 % (tools/attic/r05).
    auto stage = [&](const void* plane, const int cmin, const int ymin, const int C, const int R, unsigned char* const win, auto AUXC) {
        constexpr int AUX = decltype(AUXC)::value;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(plane), 0, (int)plane_bytes, 0x00020000);
        const unsigned magic = ((1u << 20) + (unsigned)C - 1u) / (unsigned)C;   // q / C == (q * magic) >> 20 for q < 4096, C <= 64 (q * (magic C - 2^20) < 2^20)
        const int nq = R * C;
#pragma unroll
        for (int i = 0; i < (CHUNKS + 64 * NW - 1) / (64 * NW); i++) {
            const int q0 = (i * NW + wave) * 64;                                // wave-uniform
            if (q0 < nq) {
                const unsigned q = (unsigned)q0 + lane;
                const unsigned row = (q * magic) >> 20, col = q - __umul24(row, (unsigned)C);
                // (chunks past the window's end land behind it inside the window's LDS share; reads past the plane return 0)
                if (wg_in) {   // the window lies where mirrorCoordinate is the identity: a rectangle of the plane
                    const unsigned off = __umul24((unsigned)(ymin - kExtY) + row, pitch_b) + ((unsigned)cmin + col) * 16u - (unsigned)(kExtX * SZ);
                    HF_DBG_CHECK(q >= (unsigned)nq || (size_t)off + 16 <= (size_t)plane_bytes, 212);   // (only the padding chunks behind the window may lie past the plane: they read as 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(win + (size_t)q0 * 16), 16, off, 0, 0, AUX);
                } else {       // a tile at the frame edge: rows fold back as whole rows; a chunk that touches the left / right mirror zone is
                               // gathered element by element (a reflected run is reversed) -- once per period, not once per output and row
                    const int y = mirror_warp_bl(ymin - kExtY + (int)row, dim_y);
                    const int x0 = (cmin + (int)col) * VEC - kExtX;            // first element of the chunk in its (extended) row
                    const unsigned rowoff = __umul24((unsigned)y, pitch_b);
                    const bool zone = x0 < 1 || (!CZ && x0 + VEC - 1 > W - 2);  // (chroma: only runs left of the right zone are staged)
                    if (!zone) {
                        HF_DBG_CHECK(q >= (unsigned)nq || (size_t)rowoff + (size_t)(x0 * SZ) + 16 <= (size_t)plane_bytes, 213);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(win + (size_t)q0 * 16), 16, rowoff + (unsigned)(x0 * SZ), 0, 0, AUX);
                    } else {
                        __attribute__((aligned(16))) E v[VEC];
#pragma unroll
                        for (int k = 0; k < VEC; k++) {
                            const int xs = mirror_warp_bl(x0 + k, W);
                            const unsigned eo = rowoff + (unsigned)((CZ ? (xs & ~1) + ((x0 + k) & 1) : xs) * SZ);
                            HF_DBG_CHECK(eo + SZ <= plane_bytes, 12);
                            if constexpr (SZ == 2) v[k] = (E)__builtin_amdgcn_raw_buffer_load_b16(rsrc, eo, 0, 0);
                            else v[k] = (E)__builtin_amdgcn_raw_buffer_load_b8(rsrc, eo, 0, 0);
                        }
                        *(uint4*)(win + (size_t)q * 16) = *(const uint4*)v;
                    }
                }
            }
        }
    };
    if (need_a) stage(A, cmin_a, ymin_a, C_a, R_a, lds, std::integral_constant<int, 2>{});
    if (need_b) stage(B, cmin_b, ymin_b, C_b, R_b, lds + CHUNKS * 16, std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_waitcnt(0x0F70);                                         // vmcnt(0): this wave's share of the windows is in LDS
    __syncthreads();
    if (!full) return;                                                          // a wave without a tile only helped copying

    // ---- phase C: every output from LDS.  Address of a run's first dword in its window: ((row - ymin) C - cmin) 16 + (offset & ~3)
    const Levels lv = make_levels(a.black, a.white);
    using Src = WarpSrc<E, VEC, ROWS, 1>;
    // (Dword reads at the lanes' 16-byte stride hit 8 of the 32 banks of the 4-byte read mode: SQ_LDS_BANK_CONFLICT is 72 % of the
    // kernel's LDS cycles, the LDS array 42 % busy.  Three conflict-poor ds_read_b64 from the 8-byte-aligned address below the run
    // + one v_cndmask per dword were measured SLOWER -- 698-706 vs 675 us per 16-member launch, 76.4-76.9 vs 77.5 k frames/s: the
    // conflicts hide behind the other waves, the five selects per run do not.)
    auto lds_run = [&](const unsigned char* p8, const unsigned off, const unsigned odd) {
        const uint32_t* p = (const uint32_t*)p8;
        uint32_t w[NDW + 1];
#pragma unroll
        for (int k = 0; k <= NDW; k++) w[k] = p[k];
        return run_from_dwords<E, VEC, CZ>(w, off, odd);
    };
    const unsigned rowb_a = (unsigned)C_a * 16u, rowb_b = (unsigned)C_b * 16u;
    const unsigned char* const win_a = lds - (unsigned)(ymin_a * C_a + cmin_a) * 16u;           // (pointer arithmetic only: never dereferenced below lds)
    const unsigned char* const win_b = lds + CHUNKS * 16 - (unsigned)(ymin_b * C_b + cmin_b) * 16u;
    // Output stores: BUFFER stores -- the frame's descriptor in scalar registers, ONE 32-bit offset per lane for all outputs and rows (the
    // row pitch rides in the scalar offset) -- instead of 64-bit flat addresses rebuilt on the vector ALU for every store.  The loop over
    // the outputs is NOT unrolled: the run table in LDS made the loop body independent of j (round 3 indexed register arrays with it), and
    // five copies of it were most of the kernel's 60 KB of code.
    const unsigned out_off = (unsigned)(((size_t)CZ * H * So + (size_t)cy0 * So + cx0) * SZ);
    const unsigned out_pitch = (unsigned)So * (unsigned)SZ;
    const int out_bytes = (int)((size_t)(H + (H >> 1)) * So * SZ);
#pragma unroll 1
    for (int j = 0; j < n; j++) {
        {
            Src S;
            const uint2 d = sh.tab[j][ci];
            if (need_a) {
                const uint32_t w = base + d.x, wb = w & 0xFFFFu, odd = CZ ? (wb & 1u) : 0u, off = CZ ? (wb & ~1u) : wb;
                const unsigned char* p = win_a + __umul24(w >> 16, rowb_a) + (off & ~3u);
                HF_DBG_CHECK(p >= lds && p + (ROWS - 1) * rowb_a + 4 * NDW + 4 <= lds + CHUNKS * 16, 13);
#pragma unroll
                for (int r = 0; r < ROWS; r++) S.ra[r][0] = lds_run(p + (unsigned)r * rowb_a, off, odd);
            }
            if (need_b) {
                const uint32_t w = base + d.y, wb = w & 0xFFFFu, odd = CZ ? (wb & 1u) : 0u, off = CZ ? (wb & ~1u) : wb;
                const unsigned char* p = win_b + __umul24(w >> 16, rowb_b) + (off & ~3u);
                HF_DBG_CHECK(p >= lds + CHUNKS * 16 && p + (ROWS - 1) * rowb_b + 4 * NDW + 4 <= lds + 2 * CHUNKS * 16, 14);
#pragma unroll
                for (int r = 0; r < ROWS; r++) S.rb[r][0] = lds_run(p + (unsigned)r * rowb_b, off, odd);
            }
            const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(a.outv[j], 0, out_bytes, 0x00020000);
            warp_finish_to<E, VEC, ROWS, MODE, CZ, 16>(S, a.s12v[j], a.s21v[j], ROWS, lv, [&](const int r, const E* v) {
                __builtin_amdgcn_raw_buffer_store_b128(*(const buf_v4*)v, rsrc_out, out_off, (unsigned)r * out_pitch, 2 /* nt: streaming (plain, sc0 and sc1 stores: -9 %; sc1 nt -2.5 %; sc0 nt the same) */);
            });
        }
    }
}

// Deferred phase-plane build (hf_phase_plane.h): the chain of the NEXT source period needs the full plane of frame21, and this
// launch reads that frame anyway -- so the launch also carries plane-building workgroups (the stand-alone kernel's own task code,
// 16-byte loads and stores) for every member that asks (WarpArgs::plane21), placed right in front of the warp workgroups of the same
// picture rows: the frame's rows are fetched from HBM once for both, and the stand-alone plane kernel with its 25 MB re-read of the
// frame is not launched (prep_grid_kernel supplies the grid samples the chain of THIS period needs).
static_assert(sizeof(Geom) + sizeof(WarpBatchArgs) + sizeof(PlaneOut) <= 4096, "kernel arguments of warp_wg_kernel (a launch carries at most 4 KB)");
static_assert(sizeof(WarpArgs) == 168, "hf_kernels.h kMaxWarpBatch is sized for 168-byte members");

// Block order per member: "super rows" = [plane-building blocks,] two luma block rows, then the chroma block row of the same
// picture region (a chroma block of NW stacked tiles spans twice the picture rows of a luma block), so that a region's luma and
// chroma rows pass through L2 at about the same time.
__host__ __device__ __forceinline__ int wg_super_rows(int yb, int ub) { return ub > (yb + 1) / 2 ? ub : (yb + 1) / 2; }
__host__ __device__ __forceinline__ int wg_blocks_per_member(int wpr, int yb, int ub, int plane_blocks) {
    return (wpr * 3 + plane_blocks) * wg_super_rows(yb, ub);
}

// (88 VGPRs = 5 waves per SIMD; amdgpu_waves_per_eu(6) = 80 VGPRs + 16 spilled: 77.4-77.8 vs 78.0-78.2 k frames/s -- not kept)
template <typename E, int MODE, int NW, int ROWS>
__global__ __launch_bounds__(64 * NW) void warp_wg_kernel(const Geom g, const WarpBatchArgs batch, const PlaneOut po) {
    constexpr int VEC = 16 / (int)sizeof(E);
    const int y_groups = (g.H + ROWS - 1) / ROWS;
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_windows[];   // 2 x wg_chunks(NW) x 16 bytes
    __shared__ WgShared<NW> sh;
    const int uv_groups = ((g.H >> 1) + ROWS - 1) / ROWS;
    const int wpr = (g.W + kWarpTX * VEC - 1) / (kWarpTX * VEC);
    const int y_tiles = (y_groups + kWarpTY - 1) / kWarpTY, uv_tiles = (uv_groups + kWarpTY - 1) / kWarpTY;
    const int yb = (y_tiles + NW - 1) / NW, ub = (uv_tiles + NW - 1) / NW;     // blocks of NW stacked tiles per tile column
    const int per_sr = wpr * 3 + po.blocks;
    const int n_blocks = wg_blocks_per_member(wpr, yb, ub, po.blocks);
    const int total = n_blocks * batch.n;
    const int per_band = (total + 7) >> 3;                                     // contiguous bands of units per XCD, as in warp_fast_kernel
    const int u = (int)(blockIdx.x & 7) * per_band + (int)(blockIdx.x >> 3);
    if (u >= total) return;
    const int member = (int)fastdiv((uint32_t)u, po.per_member), blk = u - member * n_blocks;
    const WarpArgs& a = batch.s[member];
    const int srow = (int)fastdiv((uint32_t)blk, po.per_sr);
    int r = blk - srow * per_sr;
    if (r < po.blocks) {   // plane-building block: the chroma rows (= luma row pairs) of this super row x groups of 4 grid columns
        if (!a.plane21) return;
        constexpr int LR = 2 * NW * kWarpTY * ROWS;                            // luma rows per super row (2 luma block rows)
        const int cgs = (g.W >> g.rs) >> 2;
        const int task = r * (64 * NW) + (int)threadIdx.x;                     // one luma row x 4 columns per thread
        const int lr = task / cgs, cg = task - lr * cgs;
        const int y = srow * LR + lr;
        if (lr >= LR || y >= g.H) return;
        if (sizeof(E) == 2 && g.rs == 3) plane_fast_task<E, 3, 1>((const E*)a.frame21, a.plane21, g.H, g.W, g.in_stride, po.pl, y >> 1, cg, y & 1);
        else plane_fast_task<E, 4, 1>((const E*)a.frame21, a.plane21, g.H, g.W, g.in_stride, po.pl, y >> 1, cg, y & 1);
        return;
    }
    r -= po.blocks;
    const int k = (int)fastdiv((uint32_t)r, po.wpr), tcol = r - k * wpr;
    const bool chroma = k == 2;
    const int brow = chroma ? srow : 2 * srow + k;
    if (brow >= (chroma ? ub : yb)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int trow = brow * NW + wave;
    const int tx0 = tcol * kWarpTX * VEC, ty0 = brow * NW * kWarpTY * ROWS;   // origin of the workgroup's tile in its plane
    const int cx0 = tx0 + (lane & (kWarpTX - 1)) * VEC;
    const int rg = trow * kWarpTY + (lane / kWarpTX);
    const bool lane_valid = trow < (chroma ? uv_tiles : y_tiles) && cx0 < g.W && rg < (chroma ? uv_groups : y_groups);
    if (chroma) warp_wg_body<E, MODE, 1, NW, ROWS>(g, a, tx0, ty0, lane_valid, wave, wg_windows, sh, po.counters);
    else warp_wg_body<E, MODE, 0, NW, ROWS>(g, a, tx0, ty0, lane_valid, wave, wg_windows, sh, po.counters);
}

template <typename E, int VEC, bool ALIGNED>
__global__ __launch_bounds__(256) void copy_kernel(const Geom g, const E* __restrict__ src, E* __restrict__ dst,
                                                    float black, float white) {
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int cx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * VEC;
    const int rows_total = g.H + (g.H >> 1);
    if (row >= rows_total || cx0 >= g.W) return;
    const int cz = row >= g.H;
    const Levels lv = make_levels(black, white);
    const E* __restrict__ s = src + (size_t)row * g.in_stride + cx0;   // UV plane starts at row H
    E* __restrict__ d = dst + (size_t)row * g.out_stride + cx0;
    if (ALIGNED && cx0 + VEC <= g.W) {
        static_assert(sizeof(E) * VEC == 16, "one 16-byte load/store per thread");
        __attribute__((aligned(16))) E v[VEC];
        *(uint4*)v = *(const uint4*)s;
#pragma unroll
        for (int i = 0; i < VEC; i++) v[i] = (E)(cz ? levels_uv<E>((float)v[i], lv) : levels_y<E>((float)v[i], lv));
        store16_streaming(d, v);
    } else {
        for (int i = 0; i < VEC && cx0 + i < g.W; i++)
            d[i] = (E)(cz ? levels_uv<E>((float)s[i], lv) : levels_y<E>((float)s[i], lv));
    }
}

// hf_debug_bounds_selftest: one deliberately out-of-range "index" -- the debug build must trap on it, the product build compiles the check away
__global__ void bounds_selftest_kernel(const int limit, int* out) {
    const int i = (int)threadIdx.x + 64;
    HF_DBG_CHECK(i < limit, 999);
    out[threadIdx.x] = i;
}

__global__ void rcp_probe_kernel(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_rcpf(in[i]);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_blur_flow(const Geom& g, const BlurBatch& b, int radius, int zero_count, hipStream_t stream) {
    // the window-sum form of blur_flow_kernel<32, 4> applies (same test as in the kernel): then 32 x 32 tiles are the faster ones at every batch size
    const FlowLevel& L = b.s[0].last;
    const bool window_sums = L.tx && L.ty && L.log2w == 1 && !(g.lw & 1) && !(g.lh & 1) && g.lw >= 64 && g.lh >= 64 && L.nwx * 2 == g.lw && L.nwy * 2 == g.lh;
    if (radius == 4 && (b.n > 4 || window_sums)) {   // the reference's radius: 32 x 32 outputs per workgroup, taps unrolled (with the tap loops a single
                                                     // pair is faster with four times the workgroups: 4.3 vs 6.0 us)
        const dim3 grd((g.lw + 31) / 32, (g.lh + 31) / 32, b.n);
        const int T = 32 + 8;
        const size_t smem = (size_t)T * (T + 1) * sizeof(uint32_t) + 2 * (size_t)T * 32 * sizeof(int);   // 16.8 KB (odd row pitch, see the kernel)
        HF_LAUNCH("blur", (blur_flow_kernel<32, 4>), grd, dim3(256), smem, stream, b, g.lw, g.lh, radius, zero_count);
        return;
    }
    if (window_sums && radius >= 2 && radius <= 64 && !(radius & 1)) {   // any even radius in the window-sum form (blur_flow_kernel<32, 0>)
        const dim3 grd((g.lw + 31) / 32, (g.lh + 31) / 32, b.n);
        const int nw = 16 + radius;
        const size_t smem = (size_t)nw * nw * sizeof(uint32_t) + 2 * (size_t)nw * 17 * sizeof(int) + 2 * 17 * 17 * sizeof(int);   // 18 KB at radius 32, 39 KB at 64
        HF_LAUNCH("blur", (blur_flow_kernel<32, 0>), grd, dim3(256), smem, stream, b, g.lw, g.lh, radius, zero_count);
        return;
    }
    const dim3 grd((g.lw + 15) / 16, (g.lh + 15) / 16, b.n);
    const int T = 16 + 2 * radius;
    const size_t smem = (size_t)T * (T + 1) * sizeof(uint32_t) + 2 * (size_t)T * 16 * sizeof(int);
    if (smem > 48 * 1024)     // large radii (up to 64: 101 KB of the CU's 160 KB LDS) need the opt-in; the attribute is per device
        (void)hipFuncSetAttribute((const void*)blur_flow_kernel<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    HF_LAUNCH("blur", (blur_flow_kernel<16, 0>), grd, dim3(256), smem, stream, b, g.lw, g.lh, radius, zero_count);
}

void launch_pack_flow(const Geom& g, const int16_t* flow, uint32_t* packed, hipStream_t stream) {
    const int n = g.lw * g.lh;
    pack_flow_kernel<<<(n + 255) / 256, 256, 0, stream>>>(flow, packed, n);
}

// Fast-path launch for VB bytes per thread and row (all members of `b` in one launch).  Returns false when the
// shape of any member does not qualify.
// Does the fast kernel with VB bytes per thread and row apply to every member of `b`?  dw: dword-aligned source loads possible.
template <typename E, int VB>
static bool warp_fast_shape(const Geom& g, const WarpBatchArgs& b, bool& dw) {
    constexpr int VEC = VB / sizeof(E);
    const int cell = 1 << g.rs;
    const int group = cell < VEC ? cell : VEC;
    const int mode = b.s[0].mode;
    bool fast = mode >= 0 && mode <= 2 && (g.in_stride % 2) == 0 && (g.out_stride % VEC) == 0 &&
                g.W >= 2 * VEC && group >= 2 && VEC % group == 0 && VEC / group <= 4;   // (group >= 2: a chroma run is made of element PAIRS)
    // dword-aligned source loads (load_run_dw) need dword-aligned frames and rows that end on a dword
    dw = ((size_t)g.in_stride * sizeof(E)) % 4 == 0 && ((size_t)g.W * sizeof(E)) % 4 == 0 && ((size_t)g.H * g.in_stride * sizeof(E)) % 4 == 0;
    for (int m = 0; m < b.n && fast; m++) {
        const WarpArgs& a = b.s[m];
        // blend shortcuts of the fast kernel need 0 <= t <= 1 and levels that cannot produce NaN;
        // chroma runs are read with element-pair granularity: needs an even input stride
        const bool sane = a.white != a.black && a.white != 0.0f && a.white == a.white && a.black == a.black;
        fast = fast && a.mode == mode && (mode != 2 || sane) && a.flow_xy && a.n_out >= 1 && a.n_out <= kMaxWarpOutputs;
        for (int i = 0; i < a.n_out && fast; i++)
            fast = fast && a.s12v[i] >= 0.0f && a.s12v[i] <= 1.0f && (((uintptr_t)a.outv[i]) & (VB - 1)) == 0;
        dw = dw && (((uintptr_t)a.frame12 | (uintptr_t)a.frame21) & 3) == 0;
    }
    return fast;
}
template <typename E>
static constexpr bool warp_small_frame(const Geom& g) {   // frames up to 1080p 8-bit: 8 bytes per thread (twice the waves)
    return (size_t)g.W * g.H * sizeof(E) <= (size_t)1920 * 1088;
}

// Can the staged kernel build the phase planes of its members' frame21 (emit_plane_rows)?  The geometry part of the answer.
static bool plane_emission_geometry(const Geom& g, const PhaseLayout& pl) {   // = the conditions of the fast plane kernel (hf_flow.hip launch_prep_fast)
    const size_t esz = g.hdr ? 2 : 1;
    const int lw = g.W >> g.rs;
    return g.rs >= 3 && g.rs <= 4 && pl.rs == g.rs && (lw << g.rs) == g.W && lw == g.lw && (lw & 3) == 0 && pl.mx <= lw && (pl.mx & 3) == 0 &&
           (pl.lwp & 3) == 0 && ((size_t)g.in_stride * esz) % 16 == 0 && ((size_t)g.H * g.in_stride * esz) % 16 == 0 && (g.H & 1) == 0;
}

template <typename E, int VB>
static bool launch_warp_fast(const Geom& g, const WarpBatchArgs& b, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                             const PhaseLayout* pl = nullptr, bool* planes_built = nullptr) {
    constexpr int VEC = VB / sizeof(E);
    const int cell = 1 << g.rs;
    const int group = cell < VEC ? cell : VEC;
    const int mode = b.s[0].mode;
    bool dw = false;
    if (!warp_fast_shape<E, VB>(g, b, dw)) return false;
    if (t_launch_observer && !ev0 && !ev1) {   // timeline (hf_kernels.h): the dispatch carries the observer's events
        hipEvent_t o0 = nullptr, o1 = nullptr;
        if (t_launch_observer->next("warp_period", &o0, &o1)) { ev0 = o0; ev1 = o1; }
    }
    const int rows = 2;  // rows per thread (divides the 2^rs rows of a flow cell); measured on MI355X, 2160p HDR blend: 1 row 25.9 us, 2 rows 24.3 us, 4 rows 30.2 us
                         // (re-measured with the final kernel, fused period HBM-cold: 2 rows 51.1 us, 4 rows 59.3 us -- halving the
                         // per-element scalar work does not pay for halving the number of waves)
    const int y_groups = (g.H + rows - 1) / rows, uv_groups = ((g.H >> 1) + rows - 1) / rows;
    const int wpr = (g.W + kWarpTX * VEC - 1) / (kWarpTX * VEC);
    const int n_tiles = wpr * ((y_groups + kWarpTY - 1) / kWarpTY + (uv_groups + kWarpTY - 1) / kWarpTY);
    int max_out = 1;
    for (int m = 0; m < b.n; m++) max_out = b.s[m].n_out > max_out ? b.s[m].n_out : max_out;
    // outputs per thread: everything for large frames; one for frames up to 1080p (measured, fused 5-output period, us:
    // 1080p SDR 23.6 / 21.9 / 20.3 / 18.9 and 1080p HDR 19.0 / 19.1 with 6 / 3 / 2 / 1 outputs per thread; 2160p HDR
    // 45.9 / 46.6 hot, 50.8 / 52.1 HBM-cold with 6 / 1)
    // ... unless the launch has rounds of waves to spare (batched periods): then every thread produces all outputs there too
    // (1080p SDR 24 -> 60, 2 batches of 16: 107.9 -> 114.2 k frames/s)
    const bool small_frame = (size_t)g.W * g.H * sizeof(E) <= (size_t)1920 * 1088 * 2;
    const int out_chunk = small_frame && (long)n_tiles * b.n < 4 * 8192 ? 1 : kMaxWarpOutputs;
    const int n_chunks = (max_out + out_chunk - 1) / out_chunk;
    // large workgroups only where the launch keeps every CU supplied with them (>= 4 rounds of 8,192 resident waves)
    // one LDS window per workgroup of kWgWaves stacked wave tiles (warp_wg_kernel): one flow cell per 16-byte thread, all outputs of the
    // period per thread, dword-aligned frames, launches of several rounds of waves (inside the pipeline the staged launch is 10 % shorter
    // than the global path -- 1,385 vs 1,545-1,595 us per 16 members, round 3)
    constexpr int WR = kWgRows, NW = kWgWaves * 2 / WR;                        // rows per thread, waves per workgroup (tile height kWgWaves x 8 rows)
    const int y_tiles_ = (((g.H + WR - 1) / WR) + kWarpTY - 1) / kWarpTY, uv_tiles_ = ((((g.H >> 1) + WR - 1) / WR) + kWarpTY - 1) / kWarpTY;
    const int plane_blocks = ((g.lw >> 2) * (2 * NW * kWarpTY * WR) + 64 * NW - 1) / (64 * NW);   // (groups of 4 columns) x (luma rows of a super row) tasks
    // (its workgroups decode their unit index with scalar multiply-high divisions, exact while units x blocks per member < 2^32: frames
    //  far beyond 8K take the generic launch below)
    const uint32_t nb_max = (uint32_t)wg_blocks_per_member(wpr, (y_tiles_ + NW - 1) / NW, (uv_tiles_ + NW - 1) / NW, plane_blocks);
    if constexpr (VB == 16) if (group == VEC && dw && out_chunk > 1 && max_out >= 2 && (long)n_tiles * b.n >= kWgMinWaves &&
                                fastdiv_exact((uint64_t)nb_max * b.n + 8, nb_max)) {
        // deferred phase planes: members that ask for one (plane21) get it from this launch if geometry and alignment allow
        WarpBatchArgs bb = b;
        PlaneOut po{};
        po.counters = b.counters;
        bool emit = pl && plane_emission_geometry(g, *pl);
        for (int m = 0; m < bb.n; m++)
            if (!emit || (((uintptr_t)bb.s[m].frame21) & 15) != 0) bb.s[m].plane21 = nullptr;
        emit = false;
        for (int m = 0; m < bb.n; m++) emit = emit || bb.s[m].plane21 != nullptr;
        if (planes_built) for (int m = 0; m < bb.n; m++) planes_built[m] = bb.s[m].plane21 != nullptr;
        if (emit) {
            po.pl = *pl;
            po.blocks = plane_blocks;
        }
        const int nb = wg_blocks_per_member(wpr, (y_tiles_ + NW - 1) / NW, (uv_tiles_ + NW - 1) / NW, po.blocks);
        const uint64_t max_unit = (uint64_t)nb * b.n + 8;   // (units of a launch incl. the padding of its grid to a multiple of 8)
        po.per_member = make_fastdiv((uint32_t)nb, max_unit); po.per_sr = make_fastdiv((uint32_t)(wpr * 3 + po.blocks), (uint64_t)nb);
        po.wpr = make_fastdiv((uint32_t)wpr, (uint64_t)wpr * 3 + po.blocks);
        const dim3 wg(((nb * b.n + 7) / 8) * 8), wb(64 * NW);
        const size_t lds_bytes = (size_t)2 * wg_chunks(NW * WR / 2) * 16;
#define HF_WARP_WG_LAUNCH(M)                                                                                                                  \
        do {                                                                                                                                      \
            auto kern = warp_wg_kernel<E, M, NW, WR>;                                                                                                 \
            if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
            hipExtLaunchKernelGGL(kern, wg, wb, lds_bytes, stream, ev0, ev1, 0, g, bb, po);                                             \
        } while (0)
        if (mode == 0) HF_WARP_WG_LAUNCH(0);
        else if (mode == 1) HF_WARP_WG_LAUNCH(1);
        else HF_WARP_WG_LAUNCH(2);
#undef HF_WARP_WG_LAUNCH
        return true;
    }
    // large workgroups only where the launch keeps every CU supplied with them (>= 4 rounds of 8,192 resident waves)
    const int wpb = out_chunk > 1 && (long)n_tiles * n_chunks * b.n >= 4 * 8192 ? warp_max_waves(sizeof(E), group, VB) : kWarpWavesSmall;
    const int n_blocks = (n_tiles + wpb - 1) / wpb;
    const dim3 fg(((n_blocks * n_chunks * b.n + 7) / 8) * 8), fb(64 * wpb);
#define HF_WARP_FAST(G, D)                                                                   \
    do {                                                                                     \
        /* ev0/ev1 (may be null): timestamps of the dispatch itself, like rocprof's kernel trace */ \
        if (mode == 0) hipExtLaunchKernelGGL((warp_fast_kernel<E, G, 2, 0, VB, D>), fg, fb, 0, stream, ev0, ev1, 0, g, b, y_groups, out_chunk, n_chunks);      \
        else if (mode == 1) hipExtLaunchKernelGGL((warp_fast_kernel<E, G, 2, 1, VB, D>), fg, fb, 0, stream, ev0, ev1, 0, g, b, y_groups, out_chunk, n_chunks); \
        else hipExtLaunchKernelGGL((warp_fast_kernel<E, G, 2, 2, VB, D>), fg, fb, 0, stream, ev0, ev1, 0, g, b, y_groups, out_chunk, n_chunks);                  \
    } while (0)
#define HF_WARP_GROUP(D)                                  \
    do {                                                  \
        if (group == VEC) HF_WARP_FAST(VEC, D);           \
        else if (group == VEC / 2) HF_WARP_FAST(VEC / 2, D); \
        else HF_WARP_FAST(VEC / 4, D);                    \
    } while (0)
    if (dw) HF_WARP_GROUP(true);
    else HF_WARP_GROUP(false);
#undef HF_WARP_GROUP
#undef HF_WARP_FAST
    return true;
}

// frames up to 1080p 8-bit: 8 bytes per thread (twice the waves); larger frames: 16 bytes per thread
template <typename E>
static bool launch_warp_fast_any(const Geom& g, const WarpBatchArgs& b, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                                 const PhaseLayout* pl = nullptr, bool* planes_built = nullptr) {
    const bool small = warp_small_frame<E>(g);   // (1080p SDR, 5-output period: 18.9 us with 8-byte threads, 26.8 us with 16-byte ones)
    if (small && launch_warp_fast<E, 8>(g, b, stream, ev0, ev1)) return true;   // (also in a batch of 16: 114.3 k frames/s against 100.2 k with 16-byte threads)
    return launch_warp_fast<E, 16>(g, b, stream, ev0, ev1, pl, planes_built);
}

template <typename E>
static void launch_warp_t(const Geom& g, const WarpArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr int VEC = 16 / sizeof(E);  // generic kernel: 16-byte stores
    WarpBatchArgs b;
    b.n = 1; b.counters = nullptr; b.s[0] = a;
    if (launch_warp_fast_any<E>(g, b, stream, ev0, ev1)) return;
    const bool aligned = (g.out_stride % VEC) == 0 && (((uintptr_t)a.out) & 15) == 0;
    const dim3 grd((g.W + 64 * VEC - 1) / (64 * VEC), (g.H + (g.H >> 1) + 3) / 4);
    if (aligned) hipExtLaunchKernelGGL((warp_kernel<E, VEC, true>), grd, dim3(256), 0, stream, ev0, ev1, 0, g, a);
    else hipExtLaunchKernelGGL((warp_kernel<E, VEC, false>), grd, dim3(256), 0, stream, ev0, ev1, 0, g, a);
}

void launch_warp(const Geom& g, const void* frame12, const void* frame21, const int16_t* flow, const uint32_t* flow_xy,
                 void* out, float t, int mode, float black, float white, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    WarpArgs a;
    a.frame12 = frame12; a.frame21 = frame21; a.flow = flow; a.flow_xy = flow_xy; a.out = out;
    a.s12 = t; a.s21 = 1.0f - t; a.mode = mode; a.black = black; a.white = white;
    a.n_out = 1; a.s12v[0] = a.s12; a.s21v[0] = a.s21; a.outv[0] = out; a.plane21 = nullptr;
    if (g.hdr) launch_warp_t<uint16_t>(g, a, stream, ev0, ev1);
    else launch_warp_t<uint8_t>(g, a, stream, ev0, ev1);
}

// Launch arguments of members [first, first + b.n) of a set of periods; false: a member's n_out is out of range.
static bool fill_warp_batch(const WarpPeriod* periods, int n, int first, int mode, WarpBatchArgs& b) {
    b.n = n - first < kMaxWarpBatch ? n - first : kMaxWarpBatch;
    b.counters = periods[0].counters;
    for (int m = 0; m < b.n; m++) {
        const WarpPeriod& p = periods[first + m];
        if (p.n_out < 1 || p.n_out > kMaxWarpOutputs) return false;
        WarpArgs& a = b.s[m];
        a.frame12 = p.frame12; a.frame21 = p.frame21; a.flow = p.flow; a.flow_xy = p.flow_xy; a.out = p.outs[0];
        a.mode = mode; a.black = p.black; a.white = p.white;
        a.n_out = p.n_out;
        for (int i = 0; i < p.n_out; i++) { a.s12v[i] = p.ts[i]; a.s21v[i] = 1.0f - p.ts[i]; a.outv[i] = p.outs[i]; }
        a.s12 = a.s12v[0]; a.s21 = a.s21v[0];
        a.plane21 = p.plane21;
    }
    return true;
}
template <typename E>
static bool warp_fast_any_shape(const Geom& g, const WarpBatchArgs& b) {
    bool dw = false;
    return (warp_small_frame<E>(g) && warp_fast_shape<E, 8>(g, b, dw)) || warp_fast_shape<E, 16>(g, b, dw);
}

bool launch_warp_periods(const Geom& g, int n, const WarpPeriod* periods, int mode, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                         const PhaseLayout* pl, bool* planes_built) {
    if (planes_built) for (int m = 0; m < n; m++) planes_built[m] = false;
    if (n < 1 || n > kMaxFlowBatch) return false;
    // at most kMaxWarpBatch members per launch (kernel-argument space): a batch of 32 is two launches.  ALL of them are
    // checked before the first one goes out, so a set of periods is either rendered by these launches or not touched at all
    for (int first = 0; first < n; first += kMaxWarpBatch) {
        WarpBatchArgs b;
        if (!fill_warp_batch(periods, n, first, mode, b)) return false;
        if (!(g.hdr ? warp_fast_any_shape<uint16_t>(g, b) : warp_fast_any_shape<uint8_t>(g, b))) return false;
    }
    for (int first = 0; first < n; first += kMaxWarpBatch) {
        WarpBatchArgs b;
        fill_warp_batch(periods, n, first, mode, b);
        hipEvent_t e0 = first == 0 ? ev0 : nullptr, e1 = first + kMaxWarpBatch >= n ? ev1 : nullptr;
        bool* pb = planes_built ? planes_built + first : nullptr;
        const bool ok = g.hdr ? launch_warp_fast_any<uint16_t>(g, b, stream, e0, e1, pl, pb) : launch_warp_fast_any<uint8_t>(g, b, stream, e0, e1, pl, pb);
        if (!ok) return false;   // (cannot happen after the check above)
    }
    return true;
}

bool warp_period_can_build_planes(const Geom& g, const PhaseLayout& pl, int n_members) {
    const size_t esz = g.hdr ? 2 : 1;
    const int VEC = (int)(16 / esz), cell = 1 << g.rs;
    // one flow cell per 16-byte thread, no 8-byte threads.  (Frames that take warp_fast_kernel -- 1080p and smaller -- keep their eager planes:
    // plane-building workgroups in THAT launch were built and measured in round 5, bit-exact and 2 % slower than the stand-alone plane kernel
    // there: 131.3-131.7 k against 134.0-134.5 k frames/s at 1080p SDR -- the frame is small enough to be re-read from L2, and the deferred order
    // adds the grid-sample launch; tools/attic/r05/deferred_planes_fast_kernel.diff)
    if (cell < VEC || (size_t)g.W * g.H * esz <= (size_t)1920 * 1088) return false;
    const int y_groups = (g.H + 1) / 2, uv_groups = ((g.H >> 1) + 1) / 2;
    const int wpr = (g.W + kWarpTX * VEC - 1) / (kWarpTX * VEC);
    const long n_tiles = (long)wpr * ((y_groups + kWarpTY - 1) / kWarpTY + (uv_groups + kWarpTY - 1) / kWarpTY);
    const int per_launch = n_members < kMaxWarpBatch ? n_members : kMaxWarpBatch;
    return n_tiles * per_launch >= kWgMinWaves && g.H == (g.lh << g.rs) && plane_emission_geometry(g, pl);
}

template <typename E>
static void launch_copy_t(const Geom& g, const void* src, void* out, float black, float white, hipStream_t stream) {
    constexpr int VEC = 16 / sizeof(E);
    const bool aligned = (g.in_stride % VEC) == 0 && (g.out_stride % VEC) == 0 &&
                         (((uintptr_t)src | (uintptr_t)out) & 15) == 0;
    const dim3 grd((g.W + 64 * VEC - 1) / (64 * VEC), (g.H + (g.H >> 1) + 3) / 4);
    if (aligned) copy_kernel<E, VEC, true><<<grd, 256, 0, stream>>>(g, (const E*)src, (E*)out, black, white);
    else copy_kernel<E, VEC, false><<<grd, 256, 0, stream>>>(g, (const E*)src, (E*)out, black, white);
}

void launch_copy(const Geom& g, const void* src, void* out, float black, float white, hipStream_t stream) {
    if (g.hdr) launch_copy_t<uint16_t>(g, src, out, black, white, stream);
    else launch_copy_t<uint8_t>(g, src, out, black, white, stream);
}

void launch_bounds_selftest(int* scratch, hipStream_t stream) {
    const int limit = 64;                            // lanes hold 64 .. 127: every one violates "i < 64" (a kernel argument: nothing of this frame outlives the call)
    bounds_selftest_kernel<<<1, 64, 0, stream>>>(limit, scratch + 1);
}

bool dbg_bounds_read_kernels(unsigned out[5], bool reset) {
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

void launch_rcp_probe(const float* in, float* out, int n, hipStream_t stream) {
    rcp_probe_kernel<<<(n + 63) / 64, 64, 0, stream>>>(in, out, n);
}

}  // namespace hf
