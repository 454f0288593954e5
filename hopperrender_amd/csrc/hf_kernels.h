// hopperrender_amd/csrc/hf_kernels.h -- internal launch interface of the gfx950 kernels.
//
// Every launcher only ENQUEUES work on `stream` (no allocation, no synchronisation) so the
// whole flow chain can be captured into a hipGraph (cdna_hip_programming.md Guideline 9).
// Citations are relative to the reference's HopperRender/ directory.
#pragma once
#include <hip/hip_runtime.h>
#ifdef __HIPCC__
#include <hip/hip_ext.h>
#endif
#include <stdint.h>

// Device debug build (`python -m hopperrender_amd.build --debug-bounds` => -DHF_DEBUG_BOUNDS, library libhopperflow_dbg.so): every
// gather index of the kernels (frame / phase-plane / flow-table / LDS-window reads) is checked against its buffer; a violation is
// RECORDED -- count + site, block and thread of the first one, in a __device__ record of the translation unit -- and the kernel goes on
// (a record the host reads with hf_debug_bounds_violations() is more robust than a trap: a trapped queue takes the device printf
// buffer and the whole context with it).  This replaces the GPU AddressSanitizer the pool cannot offer (the
// reference's own out-of-range case is the single reflection of calcDeltaSumsKernelSDR.h:86-95).  The product build compiles the
// checks away.
#ifdef HF_DEBUG_BOUNDS
#ifdef __HIPCC__
namespace hf { namespace {
__device__ unsigned int g_dbg_bounds[5];   // [0] violations, [1..4] site / block / thread / line of the first one (one record per translation unit)
__device__ __forceinline__ void dbg_bounds_record(int site, int line) {
    if (atomicAdd(&g_dbg_bounds[0], 1u) == 0u) {
        g_dbg_bounds[1] = (unsigned)site; g_dbg_bounds[2] = (unsigned)blockIdx.x; g_dbg_bounds[3] = (unsigned)threadIdx.x; g_dbg_bounds[4] = (unsigned)line;
    }
}
} }
#define HF_DBG_CHECK(cond, site) do { if (!(cond)) ::hf::dbg_bounds_record((site), __LINE__); } while (0)
#else
#define HF_DBG_CHECK(cond, site) do { } while (0)
#endif
#else
#define HF_DBG_CHECK(cond, site) do { } while (0)
#endif

namespace hf {

// ---- per-launch device timestamps without a profiler (hf_batch_timeline_*, include/hopperflow.h) ----
// `rocprofv3 --kernel-trace` makes the four batch streams host-issue-bound (its per-dispatch cost), so its timeline is not the timed
// run's.  Instead every launch of the library can carry the start / stop events of its own dispatch (hipExtLaunchKernelGGL: the same
// timestamps the kernel trace reads): while a thread has a LaunchObserver installed, HF_LAUNCH hands each launch a pair of events from it
// and notes the kernel's name; otherwise it is the plain launch.  Installed only by hf_batch_run_period of a batch whose timeline is on.
struct LaunchObserver {
    virtual bool next(const char* kernel_name, hipEvent_t* start, hipEvent_t* stop) = 0;   // false: out of records, launch unobserved
protected:
    ~LaunchObserver() = default;
};
extern thread_local LaunchObserver* t_launch_observer;

#ifdef __HIPCC__
#define HF_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                                     \
    do {                                                                                                            \
        hipEvent_t hf_e0_ = nullptr, hf_e1_ = nullptr;                                                              \
        if (::hf::t_launch_observer && ::hf::t_launch_observer->next((NAME), &hf_e0_, &hf_e1_))                    \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, hf_e0_, hf_e1_, 0, __VA_ARGS__);                \
        else                                                                                                        \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                                      \
    } while (0)
#endif

// Geometry shared by all kernels (reference ctor, opticalFlowCalcSDR.cpp:206-222).
struct Geom {
    int hdr;             // 0: uint8 elements, 1: uint16 elements
    int H, W;            // full-resolution luma size
    int in_stride;       // elements
    int out_stride;      // elements
    int rs;              // resolution scalar
    int lw, lh;          // low-res grid
};

// Division of a wave-uniform index by a launch constant on the SCALAR unit: u / d == mulhi(u, ceil(2^32 / d)) while u * d < 2^32.
// Left to the compiler a uniform u / d is ~25 VECTOR instructions (v_rcp_iflag_f32, v_mul_hi_u32 ...) and every workgroup of the
// batched kernels decodes its unit index with three of them before it can start.  Every launcher builds its dividers with the largest
// index the launch decodes (make_fastdiv(d, max_u)): where the multiply-high form would not be exact -- grids far beyond 8K -- the
// divider carries magic == 0 and the kernel takes the plain division (a uniform branch; tests/test_fastdiv_math.py).
struct FastDiv { uint32_t d, magic; };
inline bool fastdiv_exact(uint64_t max_u, uint32_t d) { return max_u * d < (1ull << 32); }
inline FastDiv make_fastdiv(uint32_t d, uint64_t max_u) {
    return FastDiv{d, d > 1 && fastdiv_exact(max_u, d) ? (uint32_t)(((1ull << 32) + d - 1) / d) : 0u};
}
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t fastdiv(uint32_t u, const FastDiv f) { return f.magic ? __umulhi(u, f.magic) : f.d > 1 ? u / f.d : u; }
#endif

// Device-side diagnostic counters of a context (hf_debug_counters_enable): u32 [kCounterWords]
//   [0 .. 2]                       period-warp workgroups by path: staged window, interior-global, generic (warp_wg_kernel)
//   [8 + 4 k + 2 axis + {0, 1}]    level k of the chain: windows of its table tiles, those of them that reused their SAD vectors
constexpr int kCounterWarp = 0, kCounterLevels = 8, kCounterWords = 8 + 4 * 16;
constexpr int kMaxFlowBatch = 32;      // contexts per hf_batch (FlowBatch below)
constexpr int kMaxWarpBatch = 16;      // members per fused warp launch (its per-member arguments are 168 bytes; a launch carries 4 KB)
constexpr int kMaxWarpOutputs = 6;     // outputs of one source period at 24 -> 120 fps (HopperRender.cpp:944-948)

// Phase-plane layout of a frame (hf_flow.hip).  ONE plane of 4-byte elements, one element per grid column, per pair of
// luma phases and per full-resolution luma row:
//     PP[y][ph2][j] = Y[y][x] | Y[y][x + 1] << 8 | U[y >> 1][x & ~1] << 16 | V[y >> 1][x & ~1] << 24      (top 8 bits each)
//     x = (j << rs) + 2 * ph2, mirrored once at the frame edge; j in [-mx, lwp - mx).
// A candidate sample of `PX` consecutive grid pixels is PX consecutive elements: one DWORD-ALIGNED 16-byte load (luma and
// chroma together), whatever the candidate offset is.  (Round 1 kept byte planes for luma and 2-byte planes for chroma:
// their 4- and 8-byte strips started at arbitrary byte offsets, and a vector load that is not dword-aligned takes a 3-4x
// slower path through the texture addresser -- the chain kernels ran at 78 % TA busy.)
struct PhaseLayout {
    int rs, nph, nph2;       // 2^rs luma phases, max(1, nph/2) phase pairs
    int mx;                  // left margin in grid units (covers every reachable offset, reflection baked in)
    int lwp;                 // row pitch in elements (multiple of 4)
    size_t bytes;            // H * nph2 * lwp * 4
};
PhaseLayout make_phase_layout(const Geom& g, int max_iterations);

// Offsets of one refinement level: one (x, y) pair per window of size `window`.
struct FlowLevel {
    int window, log2w;       // window size (power of two >= 2)
    int nwx, nwy;            // windows per grid row / column
    int16_t* tx;             // [nwy][nwx] X offsets after this level (nullptr: level does not exist = all zero)
    int16_t* ty;             // [nwy][nwx] Y offsets after this level
};

// A large-window step whose argmin has not been taken yet: the NEXT launch resolves it in its prologue
// (every workgroup recomputes the 16-way argmin of the window it lies in; the workgroup at the window's
// origin also stores the result in the level table) instead of a separate tiny launch.  Only used for
// steps without a neighbour term (levels < 4), so nothing in the consuming launch reads the table entry.
struct PendingArgmin {
    int active;
    int axis;                // axis of the pending step
    int capture_delta;       // pending step is the first of the chain: emit m_totalFrameDelta
    int use_neighbors;       // the pending step has a neighbour term (level >= 4): only the explicit argmin launch handles it
    FlowLevel lvl;           // level of the pending step (its table receives the result)
    FlowLevel lvl_prev;      // level before it (offset the candidates were relative to); tx == nullptr: zero
    const uint32_t* sums;    // [n_windows][16] raw SAD sums of the pending step
};

// One level (or one axis of a level for windows > 32) of the refinement chain
// (calcDeltaSums + determineLowestLayer + adjustOffsetArray of the reference).
struct FlowStep {
    const uint32_t* pp1;     // frame N-1 phase plane (candidates are sampled here)
    const uint32_t* pp2;     // frame N phase plane (its phase 0 = the grid samples)
    PhaseLayout pl;
    FlowLevel cur, prev;     // prev.tx == nullptr on the first level
    uint32_t* sums;          // [n_windows][16] raw SAD sums (windows > 32 only)
    uint32_t* total_delta;   // device slot of m_totalFrameDelta
    int axis;                // windows > 32: 0 = X step, 1 = Y step
    int R;                   // search radius = candidate count (2..16)
    int use_neighbors;       // iteration >= 4 (calcDeltaSumsKernelSDR.h:3,112)
    int delta_scalar, neighbor_scalar;
    int capture_delta;       // first step of the chain: emit m_totalFrameDelta
    uint32_t delta_divisor;  // lh*lw*10 (SDR) / lh*lw*6 (HDR)
    PendingArgmin pend;      // previous large-window step still to be resolved (see above)
    // ---- SAD tables: exact reuse of candidate SADs across steps (hf_flow.hip "SAD TABLES") ----
    FlowLevel prev2;         // the level before `prev` (tx == nullptr: none, all zero): prev - prev2 = what the parent window chose
    uint32_t* sadtab;        // [2 axes][sad_nby][sad_nbx][8]: the 16 candidate SADs (u16 pairs) of every 2x2 grid block as the last step
                             // of that axis that computed them left them; nullptr: reuse off (HF_FLAG_NO_SAD_REUSE)
    int sad_nbx, sad_nby;
    int sad_read;            // the previous level's launch left valid tables for every full tile (this is not the chain's first small level)
    int sad_write;           // this launch refreshes them where it computes (windows 32 .. 4)
    int y_rows_lds;          // this launch has dynamic LDS for the candidate rows of its Y steps (ysads_tile_lds / ysads_win8_lds)
    uint32_t* still_count;   // device: windows of the 32-level that chose d = 0 on both axes in this chain (the content hint, hf_calc.hip), or nullptr
    uint32_t* counters;      // diagnostic counters (hopperflow_diag.h hf_debug_counters, layout kCounter* below), nullptr: none
    int level_index;         // k of this level (its counter slot)
    // allocation bases of the per-level tables and of the window sums: a batched launch carries member 0's FlowStep and rebases it
    int16_t* tables_base;
    uint32_t* sums_base;
};

// The refinement chains of up to kMaxFlowBatch independent frame pairs of the SAME geometry and parameters run as
// one set of launches: every kernel of the chain takes the whole batch and blockIdx.z selects the pair.  One
// 480x270 chain is 135..1080 workgroups of latency-bound work per launch -- far too little for 256 CUs -- and
// the device runs at most a handful of HW queues side by side, so a throughput driver (SURVEY.md 8(e):
// independent pairs) gets its parallelism from the batch, not from more streams.  n == 1 is the drop-in path.
struct FlowBatch {
    int n;
    FlowStep s[kMaxFlowBatch];
};
struct BlurItem {
    FlowLevel last;          // offsets of the last level (tx == nullptr: all zero)
    int16_t* blurred;        // out [2][lh][lw]
    uint32_t* packed;        // out x | y << 16 per grid point (what the fast warp kernel reads)
    uint32_t* zero;          // window sums the (last) kernel of the chain clears for the next chain, or nullptr
    uint32_t* still_count;   // FlowStep::still_count of the chain: published to *still_out (mapped host memory) and cleared, or nullptr
    uint32_t* still_out;
};
struct BlurBatch {
    int n;
    BlurItem s[kMaxFlowBatch];
};

// Re-lay a freshly uploaded frame as phase planes (once per frame).
void launch_prep_frame(const Geom& g, const PhaseLayout& pl, const void* frame, uint32_t* pp, hipStream_t stream);
// The same for n frames of one geometry in one launch (hf_batch).
struct PrepBatch {
    int n;
    const void* frame[kMaxFlowBatch];
    uint32_t* pp[kMaxFlowBatch];
};
void launch_prep_frames(const Geom& g, const PhaseLayout& pl, const PrepBatch& b, hipStream_t stream);
// Deferred plane build: only the grid samples of the new frames (phase pair 0 of the rows cy << rs -- all the chain reads of the
// NEWER frame's plane); the full plane follows from the next period's warp launch (launch_warp_periods) or launch_prep_frames.
void launch_prep_grid(const Geom& g, const PhaseLayout& pl, const PrepBatch& b, hipStream_t stream);
// Windows <= 32: X and Y step of one level in a single launch.
void launch_flow_level_small(const Geom& g, const FlowBatch& b, hipStream_t stream);
// Windows > 32, one axis: partial SAD sums (atomics), then argmin + table update.
void launch_flow_big_partial(const Geom& g, const FlowBatch& b, hipStream_t stream);
void launch_flow_big_argmin(const Geom& g, const FlowBatch& b, hipStream_t stream);
// m_offsetArray view ([2][lh][lw] int16) of the last level, for parity taps.
void launch_expand_offsets(const Geom& g, const FlowLevel& last, int16_t* out, hipStream_t stream);
// blurFlowKernel with a runtime radius (4 == reference); in: offsets of the last level, out: [2][lh][lw].
// zero_count: elements of BlurItem::zero to clear.
void launch_blur_flow(const Geom& g, const BlurBatch& b, int radius, int zero_count, hipStream_t stream);
void launch_pack_flow(const Geom& g, const int16_t* flow, uint32_t* packed, hipStream_t stream);
// All outputs of one source period in ONE launch (fast path only: modes 0-2 etc.), for up to kMaxFlowBatch contexts of
// the same geometry at once; returns false when the shape of a member does not qualify and the caller must fall back to
// one launch_warp per output.
struct WarpPeriod {
    const void* frame12;
    const void* frame21;
    const int16_t* flow;
    const uint32_t* flow_xy;
    int n_out;
    void* outs[kMaxWarpOutputs];
    float ts[kMaxWarpOutputs];
    float black, white;      // already scaled for HDR
    uint32_t* plane21 = nullptr;   // deferred phase plane: build the full plane of frame21 here if the launch can (see below)
    uint32_t* counters = nullptr;  // diagnostic counters of the launch (member 0's are used), nullptr: none
};
// pl + planes_built[n] (optional): members whose period carries a plane21 pointer get the full phase plane of their frame21 built
// by the launch itself when it is the workgroup-staged kernel and geometry / alignment allow (planes_built[m] says so); everyone
// else's plane stays untouched and has to be built by launch_prep_frames.
bool launch_warp_periods(const Geom& g, int n, const WarpPeriod* periods, int mode, hipStream_t stream,
                         hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const PhaseLayout* pl = nullptr,
                         bool* planes_built = nullptr);
// Static part of that decision for a batch of n_members contexts of geometry g (hf_batch decides once whether it defers its planes).
bool warp_period_can_build_planes(const Geom& g, const PhaseLayout& pl, int n_members);
// warpFrameKernel, both planes in one launch.  black/white already scaled for HDR.
void launch_warp(const Geom& g, const void* frame12, const void* frame21, const int16_t* flow, const uint32_t* flow_xy,
                 void* out, float t, int mode, float black, float white, hipStream_t stream,
                 hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);  // events: timestamps of the dispatch itself
// copyFrameKernel, both planes in one launch.
void launch_copy(const Geom& g, const void* src, void* out, float black, float white, hipStream_t stream);
// Debug-bounds records of the two kernel translation units (hf_kernels.hip / hf_flow.hip): out[0] += violations, out[1..4] = the first
// one's site / block / thread / source line; reset: zero the records.  Return false in a build without HF_DEBUG_BOUNDS.
bool dbg_bounds_read_kernels(unsigned out[5], bool reset);
bool dbg_bounds_read_flow(unsigned out[5], bool reset);
// Self-test: a kernel with 64 out-of-range "indices" (scratch: >= 65 ints of device memory): a checking build records 64 violations of site 999.
void launch_bounds_selftest(int* scratch, hipStream_t stream);
// v_rcp_f32 of the device (parity tooling: the reference's levels use it through OpenCL's fdiv).
void launch_rcp_probe(const float* in, float* out, int n, hipStream_t stream);

}  // namespace hf
