/* oracle/hf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's OpticalFlowCalc{SDR,HDR} hot path
 * (HopperLogger/HopperRender V2.0.2.9).  It exists to CHECK the HIP product path; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (hopperrender_amd/, include/) never links, imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit against outputs of the
 * reference itself (oracle/_ref, the unmodified reference host code + OpenCL kernels run on
 * an MI355X) -- see tests/golden/ and tests/test_oracle_golden.py.  The reference ships no
 * tests or golden vectors of its own.
 *
 * All file:line citations are relative to /root/reference/HopperRender/.
 */
#ifndef HF_ORACLE_H
#define HF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Frame geometry, as derived by the reference constructor (opticalFlowCalcSDR.cpp:206-222). */
typedef struct hfo_geom {
    int hdr;        /* 0: NV12, 8-bit elements; 1: P010, 16-bit elements                       */
    int H, W;       /* full-resolution luma size                                               */
    int in_stride;  /* ELEMENTS per row of input frames  (<=0 -> W, opticalFlowCalcSDR.cpp:212) */
    int out_stride; /* ELEMENTS per row of output frames (<=0 -> W, :213)                       */
    int rs;         /* resolution scalar: smallest shift with H >> rs <= maxCalcRes (:217-220)  */
    int lw, lh;     /* low-res grid = ceil(W / 2^rs), ceil(H / 2^rs) (:221-222)                 */
} hfo_geom;

/* Counters the oracle keeps so tests can tell when a case left the reference's DEFINED
 * behaviour (the reference's single-reflection mirror can index outside the frame for
 * offsets larger than the frame, calcDeltaSumsKernelSDR.h:86-95; on a GPU that is UB). */
typedef struct hfo_diag {
    uint64_t oob_samples; /* number of cost evaluations whose mirrored position was still out of range */
} hfo_diag;

void hfo_make_geom(hfo_geom* g, int hdr, int H, int W, int in_stride, int out_stride, int max_calc_res);

/* opticalFlowCalcSDR.cpp:49-59 : initial window = nextpow2(max(lw,lh)) / 2 */
int hfo_initial_window(int lw, int lh);
/* opticalFlowCalcSDR.cpp:62-65 : iterations = requested ? min(requested, log2 ws0) : log2 ws0 */
int hfo_iterations(int ws0, int requested);
/* calcDeltaSumsKernelSDR.h:70-74 : candidate spacing sgn(d)*d*d, d = layer - R/2 */
int hfo_rel_offset(int layer, int R);

/* calcDeltaSumsKernelSDR.h:36-191 (+HDR).  sums is the reference's SPARSE layout
 * uint32 [R][lh][lw]; only window-origin entries are written (others left untouched,
 * caller zero-fills like opticalFlowCalcSDR.cpp:75-76).  Wrapping uint32 arithmetic. */
void hfo_calc_delta_sums(uint32_t* sums, const void* frame1, const void* frame2, const int16_t* offsets,
                         const hfo_geom* g, int window, int R, int iteration, int step,
                         int delta_scalar, int neighbor_scalar, hfo_diag* diag);

/* determineLowestLayerKernelSDR.h:4-28 : first-minimum argmin at window origins */
void hfo_determine_lowest_layer(const uint32_t* sums, uint8_t* lowest, int window, int R, int lh, int lw);

/* adjustOffsetArrayKernelSDR.h:4-21 */
void hfo_adjust_offsets(int16_t* offsets, const uint8_t* lowest, int window, int R, int lh, int lw, int step);

/* blurFlowKernelSDR.h:17-92 ; radius 4 == reference, other radii = the generalised formula
 * of SURVEY.md section 8 row a9 (taps [-r, r-1]^2, divide by (2r)^2, C truncation). */
void hfo_blur_flow(const int16_t* offsets, int16_t* blurred, int lh, int lw, int radius);

/* opticalFlowCalcSDR.cpp:44-116 : fill + 2*iterations x (sums, lowest, adjust) + blur.
 * frame1 = frame N-1, frame2 = frame N (:79-80).  offsets_out/blurred_out: int16 [2][lh][lw].
 * total_frame_delta: the bug-compatible m_totalFrameDelta (:91-94; HDR divisor 6,
 * opticalFlowCalcHDR.cpp:93).  iterations = 0 -> auto. */
void hfo_calculate_optical_flow(const void* frame1, const void* frame2, const hfo_geom* g, int R,
                                int iterations, int delta_scalar, int neighbor_scalar, int blur_radius,
                                int16_t* offsets_out, int16_t* blurred_out, uint32_t* total_frame_delta,
                                hfo_diag* diag);

/* warpFrameKernelSDR.h:116-184 (+HDR), both planes (the two launches of
 * opticalFlowCalcSDR.cpp:153-167).  frame12 = frame N-2, frame21 = frame N-1, flow = blurred
 * flow [2][lh][lw]; black/white are the USER levels (0..255); the HDR x256 of
 * opticalFlowCalcHDR.cpp:151-152 is applied inside.  mode 0..6 (HopperRender.h:10-18).
 * Elements of `out` the reference does not write are left untouched. */
void hfo_warp_frames(const void* frame12, const void* frame21, const int16_t* flow, void* out,
                     const hfo_geom* g, float t, int mode, float black, float white);

/* copyFrameKernelSDR.h:12-25 (+HDR), both planes (opticalFlowCalcSDR.cpp:170-183). */
void hfo_copy_frame(const void* src, void* out, const hfo_geom* g, float black, float white);

/* Floating-point flavour of the blend/levels arithmetic, used to characterise what the
 * reference's OpenCL compiler did on a given device:
 *   0 = strict IEEE fp32, no contraction, correctly rounded division (the x86 semantics)
 *   1 = blend contracted as fma(a, u, b*t)      2 = blend contracted as fma(b, t, a*u)
 * Levels: 0 = IEEE division, no contraction; 1 = what the reference's OpenCL build does on
 * gfx950: x / y -> x * rcp(y), and "q * max + mid" contracted to fma(q, max, mid). */
void hfo_set_blend_flavour(int flavour);
void hfo_set_levels_flavour(int flavour);
void hfo_set_rcp_override(int n, const float* y, const float* rcp_y);

#ifdef __cplusplus
}
#endif
#endif
