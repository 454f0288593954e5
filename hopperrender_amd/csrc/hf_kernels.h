// hopperrender_amd/csrc/hf_kernels.h -- internal launch interface of the gfx950 kernels.
//
// Every launcher only ENQUEUES work on `stream` (no allocation, no synchronisation) so the
// whole flow chain can be captured into a hipGraph (cdna_hip_programming.md Guideline 9).
// Citations are relative to the reference's HopperRender/ directory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hf {

// Geometry shared by all kernels (reference ctor, opticalFlowCalcSDR.cpp:206-222).
struct Geom {
    int hdr;             // 0: uint8 elements, 1: uint16 elements
    int H, W;            // full-resolution luma size
    int in_stride;       // elements
    int out_stride;      // elements
    int rs;              // resolution scalar
    int lw, lh;          // low-res grid
};

// One refinement step = one candidate axis at one window size
// (calcDeltaSums + determineLowestLayer + adjustOffsetArray of the reference).
struct StepArgs {
    const void* frame1;      // frame N-1, full resolution (candidates are sampled here)
    const uint32_t* grid2;   // frame N decimated to the flow grid: Y | U<<8 | V<<16 | valid<<24
    const int16_t* off_x;    // current X offsets [lh][lw]
    const int16_t* off_y;    // current Y offsets [lh][lw]
    int16_t* off_out;        // new offsets of the searched axis [lh][lw] (ping-pong partner)
    uint32_t* sums;          // [n_windows][16] window cost sums (windows larger than a workgroup)
    uint32_t* total_delta;   // device slot of m_totalFrameDelta
    int window;              // window size (power of two >= 2)
    int window_log2;
    int n_win_x;             // windows per grid row
    int R;                   // search radius = candidate count (5..16)
    int step;                // 0: search X, 1: search Y
    int use_neighbors;       // iteration >= 4 (calcDeltaSumsKernelSDR.h:3,112)
    int delta_scalar, neighbor_scalar;
    int capture_delta;       // first step of the chain: emit m_totalFrameDelta
    uint32_t delta_divisor;  // lh*lw*10 (SDR) / lh*lw*6 (HDR)
};

void launch_decimate(const Geom& g, const void* frame, uint32_t* grid, hipStream_t stream);
// Cost + window reduction; for windows <= 16 also argmin + offset update (single launch per step).
void launch_flow_step(const Geom& g, const StepArgs& a, hipStream_t stream);
// Windows > 16: argmin over the summed costs + offset update of every pixel of the window.
void launch_argmin_adjust(const Geom& g, const StepArgs& a, hipStream_t stream);
// blurFlowKernel with a runtime radius (4 == reference); in: two planes, out: [2][lh][lw].
// Also writes `packed` = x | y << 16 per grid point (what the fast warp kernel reads).
void launch_blur_flow(const Geom& g, const int16_t* off_x, const int16_t* off_y, int16_t* blurred, uint32_t* packed,
                      int radius, hipStream_t stream);
void launch_pack_flow(const Geom& g, const int16_t* flow, uint32_t* packed, hipStream_t stream);
// warpFrameKernel, both planes in one launch.  black/white already scaled for HDR.
void launch_warp(const Geom& g, const void* frame12, const void* frame21, const int16_t* flow, const uint32_t* flow_xy,
                 void* out, float t, int mode, float black, float white, hipStream_t stream);
// copyFrameKernel, both planes in one launch.
void launch_copy(const Geom& g, const void* src, void* out, float black, float white, hipStream_t stream);
// v_rcp_f32 of the device (parity tooling: the reference's levels use it through OpenCL's fdiv).
void launch_rcp_probe(const float* in, float* out, int n, hipStream_t stream);

}  // namespace hf
