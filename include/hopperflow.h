/* include/hopperflow.h -- C ABI of the MI355X-native optical-flow frame interpolator.
 *
 * This is the drop-in boundary for ONE hot path of HopperLogger/HopperRender: the
 * OpticalFlowCalc{SDR,HDR} calculator (reference HopperRender/opticalFlowCalc.h:24-138 and
 * opticalFlowCalc{SDR,HDR}.cpp).  The reference's filter talks to that path through a C++
 * class; include/opticalFlowCalc.h re-creates that class source-compatibly ON TOP of this C
 * ABI, and any other host language binds these symbols directly.
 *
 * Conventions: plain C types only; every call returns 0 on success or a negative hf_status;
 * nothing throws; hf_last_error() returns a human-readable string for the last failure of
 * that context (or of hf_create when ctx == NULL).  Strides are in ELEMENTS (uint8 for NV12,
 * uint16 for P010), exactly like the reference (opticalFlowCalcHDR.cpp:20,279-282).
 * One context = one GPU + one HIP stream; a context is not thread-safe (reference: one object
 * per filter instance); contexts are independent of each other, also
 * on the same device, which is how independent frame pairs are batched and sharded.
 *
 * All file:line citations are relative to the reference's HopperRender/ directory.
 */
#ifndef HOPPERFLOW_H
#define HOPPERFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HF_ABI_VERSION 6

typedef struct hf_ctx hf_ctx;

typedef enum hf_status {
    HF_OK = 0,
    HF_ERR_INVALID_ARGUMENT = -1, /* bad config / NULL pointer / blending scalar > 1 (opticalFlowCalcSDR.cpp:143-146) */
    HF_ERR_NO_DEVICE = -2,        /* no HIP device satisfies the request (reference: detectDevices throws, opticalFlowCalc.cpp:97-109) */
    HF_ERR_OUT_OF_MEMORY = -3,
    HF_ERR_HIP = -4,              /* a HIP runtime call failed (reference: CHECK_ERROR, opticalFlowCalc.h:15-22) */
    HF_ERR_STATE = -5             /* the object is in a state that excludes the call: a context that already belongs to a
                                     batch handed to hf_batch_create, asynchronous host I/O on a batch member */
} hf_status;

/* Frame output modes of warpFrames (reference HopperRender.h:10-18, warpFrameKernelSDR.h:133-183). */
typedef enum hf_output_mode {
    HF_MODE_WARPED_FRAME_12 = 0,
    HF_MODE_WARPED_FRAME_21 = 1,
    HF_MODE_BLENDED_FRAME = 2,
    HF_MODE_HSV_FLOW = 3,
    HF_MODE_GREY_FLOW = 4,
    HF_MODE_SIDE_BY_SIDE_1 = 5,
    HF_MODE_SIDE_BY_SIDE_2 = 6
} hf_output_mode;

#define HF_FLAG_ASYNC 0x1 /* calls only enqueue; results/timings valid after hf_sync(). Default: every
                             call blocks like the reference (CL_TRUE transfers, clWaitForEvents). */
#define HF_FLAG_NO_GRAPH 0x2 /* launch the flow chain eagerly instead of replaying a hipGraph (debug) */
#define HF_FLAG_PROFILE 0x4  /* bracket every warp/copy launch and every flow chain with HIP events on ctx's
                                stream; totals are read with hf_get_profile() (bench.py's live roofline figure) */
#define HF_FLAG_NO_LAZY_ARGMIN 0x8 /* large windows: take every argmin in a launch of its own (debug / A-B timing) */
#define HF_FLAG_DUAL_STREAM 0x40 /* async hosts: warp kernels on a second stream of the context.  warpFrames consumes the
                                    PREVIOUS flow and frames N-2/N-1 (opticalFlowCalcSDR.cpp:154-156) while
                                    calculateOpticalFlow produces the next flow from N-1/N into the other buffer, so the two
                                    overlap inside one context; events keep every other ordering intact */
#define HF_FLAG_NO_FUSED_WARP 0x80 /* hf_interpolate_period: one warp launch per output frame (debug / A-B timing) */
#define HF_FLAG_BATCH_NORMAL_PRIORITY 0x800 /* on the leader passed to hf_batch_create: give the batch a normal-priority stream
                                                (default: highest priority, see hf_batch_create) */
#define HF_FLAG_BATCH_EAGER_PLANES 0x1000 /* on the leader passed to hf_batch_create: hf_batch_run_period builds every phase plane in
                                              full when its frame arrives (default for large frames: grid samples only, the full
                                              plane comes out of the next period's warp launch -- see hf_batch_run_period) */
#define HF_FLAG_NO_SAD_REUSE 0x2000 /* flow chain: recompute every candidate SAD at every step, as the reference does, instead of summing the per-block SAD
                                     * tables of the last step that sampled the same positions (same results; debug / A-B timing) */
#define HF_FLAG_SAD_REUSE_ALWAYS 0x4000 /* flow chain: keep the SAD tables whatever the content (default: the chain reports how many windows kept their
                                         * offsets and the next chains drop the tables while hardly any does -- same results, other kernels) */
#define HF_FLAG_NO_TIMING 0x200 /* do not record the events behind m_ofcCalcTime / m_warpCalcTime (hf_stats times stay 0).
                                   Every timing event is a barrier packet on the stream: measured 5-6 us each between
                                   back-to-back kernels, ~15 us per source period in a throughput pipeline */
/* (0x10, 0x20, 0x100, 0x400 were round-1 stream-topology experiments -- shared warp stream, priority streams, warp
 *  turnstile, deferred phase planes -- all measured slower or equal; removed, findings in DESIGN.md section 4) */

/* The nine constructor arguments of OpticalFlowCalcSDR/HDR (opticalFlowCalcSDR.cpp:206-208)
 * plus build-side extensions (0 selects the reference behaviour for each). */
typedef struct hf_config {
    uint32_t struct_size;   /* = sizeof(hf_config), for ABI evolution */
    int32_t is_hdr;         /* 0: NV12 8-bit (OpticalFlowCalcSDR), 1: P010 16-bit (OpticalFlowCalcHDR) */
    int32_t frame_height;   /* ctor arg 1 */
    int32_t frame_width;    /* ctor arg 2 */
    int32_t input_stride;   /* ctor arg 3, elements; <= 0 -> frame_width (:212) */
    int32_t output_stride;  /* ctor arg 4, elements; <= 0 -> frame_width (:213) */
    int32_t delta_scalar;   /* ctor arg 5, config.h:23 default 8 */
    int32_t neighbor_scalar;/* ctor arg 6, config.h:24 default 6 */
    float black_level;      /* ctor arg 7, 0..255 */
    float white_level;      /* ctor arg 8, 0..255 */
    int32_t max_calc_res;   /* ctor arg 9, config.h:4 default 270 */
    /* --- extensions --- */
    int32_t device_index;   /* HIP device ordinal, or -1 = the first suitable device like the reference's detectDevices
                               (opticalFlowCalc.cpp:67-93); hf_get_device() tells which one it became */
    int32_t iterations;     /* NUM_ITERATIONS (config.h:6) made runtime; 0 = auto */
    int32_t blur_radius;    /* KERNEL_RADIUS (blurFlowKernelSDR.h:4) made runtime; 0 -> 4 */
    int32_t search_radius;  /* initial m_opticalFlowSearchRadius; 0 -> MIN_SEARCH_RADIUS 5 (:216) */
    uint32_t flags;         /* HF_FLAG_* */
} hf_config;

/* The public fields the filter reads/writes on the calculator (opticalFlowCalc.h:27-48). */
typedef struct hf_params {
    int32_t delta_scalar;    /* m_deltaScalar           (r/w by the settings thread, HopperRender.cpp:1385-1390) */
    int32_t neighbor_scalar; /* m_neighborBiasScalar */
    float black_level;       /* m_outputBlackLevel */
    float white_level;       /* m_outputWhiteLevel */
    int32_t search_radius;   /* m_opticalFlowSearchRadius, 5..16 (governor, HopperRender.cpp:1438-1463) */
    uint32_t frame_count;    /* m_frameCount (NewSegment zeroes it, HopperRender.cpp:840) */
} hf_params;

typedef struct hf_stats {
    uint32_t total_frame_delta; /* m_totalFrameDelta (bug-compatible, opticalFlowCalcSDR.cpp:91-94) */
    uint32_t frame_count;       /* m_frameCount */
    double ofc_calc_time;       /* m_ofcCalcTime: upload start -> blur end, seconds (:125-127) */
    double ofc_avg_calc_time;   /* m_ofcAvgCalcTime (:128-133) */
    double ofc_peak_calc_time;  /* m_ofcPeakCalcTime (:131,136-138) */
    double warp_calc_time;      /* m_warpCalcTime: first warp/copy launch -> readback end (:36-41) */
    int32_t res_scalar;         /* m_opticalFlowResScalar */
    int32_t low_width;          /* m_opticalFlowFrameWidth */
    int32_t low_height;         /* m_opticalFlowFrameHeight */
    int32_t frame_width, frame_height, input_stride, output_stride;
    int32_t iterations;         /* effective refinement iterations of the last flow calc */
    int32_t initial_window;     /* first window size (opticalFlowCalcSDR.cpp:49-59) */
    uint64_t input_frame_bytes; /* bytes updateFrame reads  = bpp*(H*S_in + (H/2)*S_in)  (:20) */
    uint64_t output_frame_bytes;/* bytes downloadFrame writes = bpp*(H*S_out + (H/2)*S_out) (:33) */
    uint64_t phase_plane_bytes; /* bytes of one phase plane (this build's re-laid copy of a frame) */
    int32_t sad_tables;         /* 1: the last flow calc kept SAD tables (exact cross-step reuse, csrc/hf_flow.hip), 0: it recomputed every step */
    float still_share;          /* smoothed share of 32-windows that chose d = 0 on both axes in the last chains (what switches the tables off on
                                 * content where hardly any window keeps its offsets; < 0: no report yet) */
} hf_stats;

/* ---- lifecycle: constructor / destructor (opticalFlowCalcSDR.cpp:206-325, :185-204) ---- */
int hf_create(const hf_config* cfg, hf_ctx** out_ctx);
void hf_destroy(hf_ctx* ctx);
const char* hf_last_error(const hf_ctx* ctx);
int hf_abi_version(void);

/* detectDevices (opticalFlowCalc.cpp:45-109) as a pure function over a capability table, so that the rule can be tested
 * without hardware: index of the FIRST entry with vram_bytes >= required_vram_bytes, >= 2048 bytes of LDS per workgroup,
 * workgroups of 256 threads (16 x 16) and 64-wide wavefronts, or -1.  why_not (optional) receives the reference's messages
 * for the last entry inspected (:98-108).  hf_create(device_index = -1) applies it to the HIP devices (vram = total device
 * memory, as the reference compares CL_DEVICE_GLOBAL_MEM_SIZE) and then insists on that much FREE memory, moving on if not. */
typedef struct hf_device_caps {
    uint64_t vram_bytes;
    uint64_t lds_bytes_per_workgroup;
    int32_t max_threads_per_workgroup;
    int32_t wavefront_size;
} hf_device_caps;
int hf_select_device(const hf_device_caps* caps, int n, uint64_t required_vram_bytes, char* why_not, size_t why_not_size);
int hf_get_device(const hf_ctx* ctx);   /* the HIP ordinal the context lives on */

/* ---- the five virtuals of OpticalFlowCalc (opticalFlowCalc.h:100-132) ---- */
/* updateFrame (opticalFlowCalcSDR.cpp:19-29): upload one NV12/P010 frame, rotate the 3-frame ring, frame_count++ */
int hf_update_frame(hf_ctx* ctx, const void* host_frame);
/* calculateOpticalFlow (opticalFlowCalcSDR.cpp:44-139): flow between ring[1] (N-1) and ring[2] (N) */
int hf_calculate_optical_flow(hf_ctx* ctx);
/* warpFrames (opticalFlowCalcSDR.cpp:141-168): blending_scalar in [.., 1], mode = hf_output_mode */
int hf_warp_frames(hf_ctx* ctx, float blending_scalar, int frame_output_mode);
/* copyFrame (opticalFlowCalcSDR.cpp:170-183) */
int hf_copy_frame(hf_ctx* ctx);
/* downloadFrame (opticalFlowCalcSDR.cpp:31-42): blocking readback of the output frame */
int hf_download_frame(hf_ctx* ctx, void* host_out);

/* ---- public-field access ---- */
int hf_get_params(const hf_ctx* ctx, hf_params* out);
int hf_set_params(hf_ctx* ctx, const hf_params* in);
int hf_get_stats(hf_ctx* ctx, hf_stats* out);

/* ---- device-resident variants (batch driver / benchmarks: inputs and outputs stay in HBM) ---- */
/* Same as hf_update_frame but the source is a device pointer on ctx's device (device-to-device). */
int hf_update_frame_device(hf_ctx* ctx, const void* device_frame);
/* Asynchronous host I/O on side streams of the context (page-locked buffers, e.g. hf_host_malloc_pinned): the
 * upload of frame N+1 and the readback of finished output frames overlap the flow chain and the warps; nothing
 * blocks, hf_sync() waits for everything.  The host buffers must stay valid (and, for the readback, unread) until
 * hf_sync().  Output frames rotate through an internal ring of three device buffers so that the next warp does
 * not wait for the previous readback. */
int hf_update_frame_async(hf_ctx* ctx, const void* pinned_host_frame);
int hf_download_frame_async(hf_ctx* ctx, void* pinned_host_out);
/* Streaming hosts (the multi-GPU driver, hopperrender_amd/hostio.py): what the host must know WITHOUT draining the side streams.
 *   hf_wait_flow      blocks until the last hf_calculate_optical_flow of an asynchronous context has finished and makes its
 *                     m_totalFrameDelta visible in hf_get_stats -- the filter decides warp vs copy from it (HopperRender.cpp:
 *                     959-972,1126-1176) while uploads and readbacks keep running;
 *   hf_wait_download  blocks until the index-th hf_download_frame_async of this context (0, 1, 2 ... in issue order;
 *                     hf_downloads_issued() = how many there are) has landed in its host buffer. */
int hf_wait_flow(hf_ctx* ctx);
uint64_t hf_downloads_issued(const hf_ctx* ctx);
int hf_wait_download(hf_ctx* ctx, uint64_t index);
/* Zero-copy variant: the ring keeps a REFERENCE to device_frame (e.g. a decoder surface).  The caller must
 * leave the frame untouched until three further frames have been submitted (it stays in the 3-frame ring
 * as frame N, N-1 and N-2, opticalFlowCalcSDR.cpp:22-28). */
int hf_update_frame_device_ref(hf_ctx* ctx, const void* device_frame);
/* One source-frame period in a single call (batch driver): hf_update_frame_device_ref(device_frame) if it is
 * non-NULL, hf_calculate_optical_flow(), then for i < n_out: hf_warp_frames(t[i], mode) written to
 * device_out[i].  Same results as the individual calls; only the host overhead differs. */
int hf_interpolate_period(hf_ctx* ctx, const void* device_frame, int n_out, const float* t, void* const* device_out, int mode);
/* update_and_flow = 0: only the warps of the period (no updateFrame, no calculateOpticalFlow). */
int hf_interpolate_period_ex(hf_ctx* ctx, const void* device_frame, int n_out, const float* t, void* const* device_out, int mode,
                             int update_and_flow);
/* ---- Batched flow calculation (throughput drivers) -------------------------------------------------------------
 * The reference computes one pair at a time (opticalFlowCalcSDR.cpp:44-139: 66 enqueues per call).  Its flow grid is
 * at most 480x270, so one refinement chain is a sequence of small latency-bound launches that cannot fill 256 CUs.
 * A driver that converts INDEPENDENT frame pairs groups up to 32 contexts of identical
 * geometry/parameters into a batch: hf_batch_calculate_optical_flow() runs the calculateOpticalFlow() of every
 * member as ONE set of launches (each kernel handles all pairs).  Results per member are bit-identical to
 * hf_calculate_optical_flow(member).
 *   - members: HF_FLAG_ASYNC contexts without async host I/O, all single-stream or all HF_FLAG_DUAL_STREAM, same device, frame geometry, iterations, blur radius; at call time the same search
 *     radius / delta / neighbor scalar.  DUAL_STREAM members issue their warps on up to 3 streams the batch shares
 *     out round robin (they overlap the batched chain; measured slower than single-stream batches).
 *   - the batch issues on a stream of its own of the HIGHEST priority.  Not for the priority: the HIP runtime keeps one pool of
 *     hardware queues per priority and nobody else creates such streams, so the batch streams of a process land on different
 *     hardware queues whatever was created before them (DESIGN.md "Streams and hardware queues").  SIDE EFFECT: the priority is
 *     real -- batch launches are scheduled ahead of every normal-priority stream of the process (other contexts' work and
 *     asynchronous I/O, RCCL).  A host that mixes batches with latency-sensitive single contexts passes
 *     HF_FLAG_BATCH_NORMAL_PRIORITY on the leader; if the priority range cannot be queried the batch falls back to a normal stream.
 *   - while the batch exists all members issue on ONE stream (the batch's own): their hf_update_frame_device*,
 *     hf_interpolate_period_ex(..., update_and_flow = 0), hf_sync ... calls keep working and stay in program order
 *     with the batched chain.  Destroy the batch before its members. */
typedef struct hf_batch hf_batch;
int hf_batch_create(hf_ctx* const* members, int n, hf_batch** out_batch);
void hf_batch_destroy(hf_batch* batch);
int hf_batch_calculate_optical_flow(hf_batch* batch);
/* hf_update_frame_device_ref(member i, device_frames[i]) for every member, the phase planes of all new frames built by
 * ONE launch. */
int hf_batch_update_frames_device_ref(hf_batch* batch, const void* const* device_frames);
/* hf_interpolate_period_ex(member i, NULL, n_out[i], t + 6 i, device_out + 6 i, mode, 0) for every member: the warps of
 * one source period of EVERY member in ONE launch (single-stream members, modes 0-2, every n_out[i] >= 1; otherwise
 * member by member).  t and device_out are [batch size][HF_MAX_PERIOD_OUTPUTS] arrays; a NULL device_out entry
 * selects the member's internal output frame. */
#define HF_MAX_PERIOD_OUTPUTS 6
int hf_batch_interpolate_period(hf_batch* batch, const int* n_out, const float* t, void* const* device_out, int mode);
/* One source period of the whole batch in ONE call (what a throughput driver issues per period -- three calls' worth of
 * argument marshalling matter when a period is ~70 us of GPU time): hf_batch_update_frames_device_ref(device_frames) unless
 * device_frames == NULL, hf_batch_calculate_optical_flow() if calculate_flow, hf_batch_interpolate_period(n_out, t,
 * device_out, mode) unless n_out == NULL.  Same results and same error behaviour as the three calls.
 * Where the batched period warp is the workgroup-staged kernel (batches of frames > 1080p, one flow cell per 16-byte thread) the
 * call defers the phase planes: at update time only the grid samples of the new frame are taken, and the full plane of frame
 * N-1, which the chain of the NEXT period reads, is built by that period's warp launch -- it reads the frame anyway -- which
 * is then enqueued ahead of the period's chain (it uses the previous flow and frames N-2 / N-1, so the order is free).  Saves
 * the stand-alone plane kernel's re-read of every frame.  Any period that cannot do that (no outputs, diagnostic modes, a
 * separate hf_batch_calculate_optical_flow / hf_calculate_optical_flow call) builds the missing plane with the stand-alone
 * kernel first.  Side effect: m_ofcCalcTime of the members then includes the warp launch.  HF_FLAG_BATCH_EAGER_PLANES on the
 * leader turns it off. */
int hf_batch_run_period(hf_batch* batch, const void* const* device_frames, int calculate_flow, const int* n_out, const float* t,
                        void* const* device_out, int mode);
/* 1: hf_batch_run_period defers the phase planes of this batch (see above); 0: it builds them eagerly. */
int hf_batch_defers_planes(const hf_batch* batch);
int hf_batch_sync(hf_batch* batch);   /* hf_sync() of every member */
int hf_batch_size(const hf_batch* batch);
const char* hf_batch_last_error(const hf_batch* batch);   /* batch == NULL: error of the last failed hf_batch_create (per thread) */

/* Device-to-device copy of the output frame into caller-owned device memory. */
int hf_download_frame_device(hf_ctx* ctx, void* device_out);
/* Redirect warp/copy output into caller-owned device memory (NULL restores the internal buffer). */
int hf_set_output_buffer(hf_ctx* ctx, void* device_out);
/* Block until everything enqueued on ctx's stream has finished; finalises timings/total_frame_delta. */
int hf_sync(hf_ctx* ctx);

/* ---- parity / debugging taps on the reference's device buffers ---- */
/* m_offsetArray: int16 [2][low_h][low_w] (raw flow of the last calc) */
int hf_read_offsets(hf_ctx* ctx, int16_t* host_out);
/* m_blurredOffsetArray[idx]: idx 0 = flow consumed by warpFrames, 1 = newest (opticalFlowCalcSDR.cpp:121-123) */
int hf_read_blurred_flow(hf_ctx* ctx, int idx, int16_t* host_out);
int hf_write_blurred_flow(hf_ctx* ctx, int idx, const int16_t* host_in);
/* This build's phase plane of ring frame ring_slot (0 = N-2, 1 = N-1, 2 = N; hf_stats.phase_plane_bytes bytes; the re-laid top-8-bit
 * copy of a frame that replaces the strided sampling of calcDeltaSumsKernelSDR.h:78-100).  *complete = 0: the
 * plane holds only its grid samples so far (deferred build, hf_batch_run_period). */
int hf_read_phase_plane(hf_ctx* ctx, int ring_slot, void* host_out, int* complete);

/* Per-kernel device time accumulated since the last hf_reset_profile() (needs HF_FLAG_PROFILE). */
typedef struct hf_profile {
    uint64_t warp_launches;   /* warp_kernel launches (one per warpFrames: both planes) */
    double warp_ms;           /* summed device time of those launches */
    uint64_t copy_launches;
    double copy_ms;
    uint64_t flow_chains;     /* calculateOpticalFlow chains (16 steps + blur) */
    double flow_ms;           /* summed device time first kernel start -> blur end */
    uint64_t warp_frames;     /* output frames produced by the counted warp launches (hf_interpolate_period fuses a period) */
} hf_profile;
int hf_get_profile(hf_ctx* ctx, hf_profile* out); /* synchronises ctx */
/* Bracket only every n-th warp/copy launch and every m-th flow chain (default 1/1): event records perturb
 * back-to-back launches, so throughput runs sample instead of bracketing everything. */
int hf_set_profile_interval(hf_ctx* ctx, int warp_every, int flow_every);
int hf_reset_profile(hf_ctx* ctx);

/* ---- caller protocol: what the reference's filter does AROUND the calculator (HopperRender.cpp:938-1197,1438-1463) ----
 * Host logic only (no GPU work of its own): how many output frames a source period gets and at which blending scalars,
 * the scene-change decision on the m_totalFrameDelta stream (warp vs copy -- it decides output pixels), the search-radius
 * governor.  A C++ host that keeps the reference's own filter code does not need these; tests/cpp/replay_filter.cpp,
 * the Python mirror (hopperrender_amd/protocol.py) and any cgo / JNI host do.  Times are 100-ns units (REFERENCE_TIME). */
typedef struct hf_filter hf_filter;
typedef struct hf_filter_config {
    uint32_t struct_size;
    int32_t scene_change_threshold;  /* m_iSceneChangeThreshold; < 0 -> DEFAULT_SCENE_CHANGE_THRESHOLD 200 (config.h:27) */
    int64_t source_frame_time;       /* m_rtSourceFrameTime, <= 0 -> 417083 = 23.976 fps (HopperRender.cpp:162) */
    int64_t target_frame_time;       /* m_rtTargetFrameTime, <= 0 -> 166667 = 60 fps (:163) */
    int32_t frame_output_mode;       /* m_iFrameOutput (hf_output_mode), BlendedFrame = 2 */
    int32_t auto_adjust;             /* run the governor in hf_filter_deliver (AUTO_SEARCH_RADIUS_ADJUST, config.h:12) */
    int32_t active;                  /* interpolation requested (m_iIntActiveState != Deactivated) */
    int32_t reserved;
} hf_filter_config;
typedef struct hf_filter_state {
    int32_t num_int_frames;          /* m_iNumIntFrames of the current source period */
    int32_t active;                  /* m_iIntActiveState == Active */
    double blending_scalar;          /* m_dBlendingScalar */
    double total_warp_duration;      /* m_dTotalWarpDuration, seconds */
    int64_t playback_frame_time;     /* m_rtCurrPlaybackFrameTime */
    uint32_t peak_scene_change_delta, peak_scene_change_delta2;  /* m_iPeakSceneChangeDelta{,2} (1-second window) */
    uint32_t frame_delta_history, scene_change_history;          /* entries in the two sliding windows */
    int32_t average_frame_delta, scene_change_delta1, scene_change_delta2;  /* of the last decision (:1139-1144) */
} hf_filter_state;
int hf_filter_create(const hf_filter_config* cfg, hf_filter** out_filter);
void hf_filter_destroy(hf_filter* filter);
int hf_filter_new_segment(hf_filter* filter, double rate);                       /* NewSegment (:834-845); the caller zeroes m_frameCount (:840) */
int hf_filter_set_playback_frame_time(hf_filter* filter, int64_t playback_frame_time);
int hf_filter_is_active(const hf_filter* filter);
int hf_filter_begin_source_frame(hf_filter* filter);                            /* :944-948, returns m_iNumIntFrames */
double hf_filter_blending_scalar(const hf_filter* filter);                      /* m_dBlendingScalar */
void hf_filter_advance_blending_scalar(hf_filter* filter);                      /* :1192-1197 */
void hf_filter_add_warp_duration(hf_filter* filter, double warp_calc_time);     /* :1189 */
int hf_filter_auto_adjust(hf_filter* filter, double ofc_calc_time, int32_t* search_radius); /* :1438-1463; returns -1/0/+1 */
int hf_filter_push_frame_delta(hf_filter* filter, uint32_t frame_count, uint32_t total_frame_delta);   /* :959-972 */
int hf_filter_detect_scene_change(hf_filter* filter, uint32_t frame_count);     /* :1126-1176; 1 = copyFrame instead of warpFrames */
int hf_filter_get_state(const hf_filter* filter, hf_filter_state* out);
/* One DeliverToRenderer (:938-1197) on a blocking context: host_out[i] receives output frame i (n = return of
 * hf_filter_begin_source_frame <= max_out), kinds[i] (optional) = 1 warp / 0 copy.  The number of outputs is unbounded in
 * principle (a slowed-down segment): call hf_filter_begin_source_frame() first to size host_out -- it only recomputes
 * m_iNumIntFrames from the state -- or retry after HF_ERR_INVALID_ARGUMENT with the *n_out buffers it then reports (nothing else
 * has been touched at that point). */
int hf_filter_deliver(hf_filter* filter, hf_ctx* ctx, const void* host_in, void* const* host_out, int max_out, int* n_out, int32_t* kinds);

/* ---- Streaming host-I/O driver of one rank + timeline planner (multi-GPU hosts; csrc/hf_hostio.cpp, plain host C++) --------
 * The reference's filter feeds the calculator from host memory with blocking transfers (opticalFlowCalcSDR.cpp:19-42) on one
 * GPU.  A throughput host converts one clip on several GPUs: rank r (= process = GPU) owns a contiguous chunk of the source
 * timeline (hf_shard_timeline: no exchange between ranks -- the warm-up frames in front of a chunk rebuild ring, previous flow
 * and scene-change history), and hf_hostio_run() streams it: pinned input ring -> hf_update_frame_async -> chain / warps ->
 * hf_download_frame_async -> pinned output ring -> sink(), output frames strictly in index order, the filter's warp-vs-copy
 * decision (hf_filter) made per period from that period's m_totalFrameDelta (one hf_wait_flow per period).  `fill` and `sink`
 * run on the calling thread; the buffers they are handed are page-locked and only valid during the call. */
typedef struct hf_timeline_chunk {
    int64_t first_period, n_periods;   /* source periods (= source frames) this rank interpolates */
    int64_t first_frame, n_frames;     /* source frames it has to be fed: the warm-up frames + its own */
    int64_t first_output, n_outputs;   /* global index of its first output frame, number of its output frames */
    double blend_at_start;             /* m_dBlendingScalar when its first period begins */
} hf_timeline_chunk;
/* n_out (optional): [n_periods] outputs per owned period; t (optional): their blending scalars, t_capacity entries at least
 * n_outputs (call once with n_out = t = NULL to learn the sizes).  overlap = 3 and delta_history = 12 reproduce the sequential
 * filter (delta_history 0: scene-change detection off). */
int hf_shard_timeline(int64_t n_source_frames, int world, int rank, int64_t source_frame_time, int64_t target_frame_time, int overlap,
                      int delta_history, hf_timeline_chunk* out, int32_t* n_out, float* t, int64_t t_capacity);
typedef struct hf_hostio hf_hostio;
typedef struct hf_hostio_config {
    uint32_t struct_size;
    int32_t in_ring, out_ring;          /* page-locked input / output frame buffers; <= 0 -> 3 / 12 (>= 3 / >= 2) */
    int32_t frame_output_mode;          /* m_iFrameOutput, BlendedFrame = 2 */
    int32_t scene_change_threshold;     /* < 0 -> DEFAULT_SCENE_CHANGE_THRESHOLD */
    int32_t reserved;
    int64_t source_frame_time, target_frame_time;   /* 100-ns units; <= 0 -> 417083 / 166667 */
} hf_hostio_config;
typedef int (*hf_hostio_fill_fn)(void* user, int64_t source_frame_index, void* pinned_frame);              /* 0 = ok */
typedef int (*hf_hostio_sink_fn)(void* user, int64_t output_index_in_chunk, const void* frame, int32_t kind);   /* kind: 1 warp, 0 copy */
/* ctx: an HF_FLAG_ASYNC (| HF_FLAG_DUAL_STREAM) context that the driver uses exclusively while it exists.
 * cfg == NULL: the filter's defaults -- rings 3 / 12, BlendedFrame output, DEFAULT_SCENE_CHANGE_THRESHOLD, 23.976 -> 60 fps.  (In a
 * caller-supplied struct a scene_change_threshold of 0 means what it says: every non-zero frame delta is a scene change.) */
int hf_hostio_create(hf_ctx* ctx, const hf_hostio_config* cfg, hf_hostio** out);
void hf_hostio_destroy(hf_hostio* io);
/* kinds (optional): [chunk->n_outputs] 1 warp / 0 copy per output frame. */
int hf_hostio_run(hf_hostio* io, const hf_timeline_chunk* chunk, const int32_t* n_out, const float* t, hf_hostio_fill_fn fill,
                  hf_hostio_sink_fn sink, void* user, int32_t* kinds);
int hf_hostio_get_traffic(const hf_hostio* io, uint64_t* bytes_in, uint64_t* bytes_out);   /* host -> device / device -> host so far */
const char* hf_hostio_last_error(const hf_hostio* io);   /* io == NULL: last failed hf_hostio_create / hf_shard_timeline of this thread */

/* ---- plain device-memory helpers so non-HIP hosts (ctypes, cgo, JNI) can stage frames ---- */
int hf_device_count(void);
int hf_device_malloc(int device_index, size_t bytes, void** out_dev_ptr);
int hf_device_free(int device_index, void* dev_ptr);
int hf_memcpy_h2d(int device_index, void* dev_dst, const void* host_src, size_t bytes);
int hf_memcpy_d2h(int device_index, void* host_dst, const void* dev_src, size_t bytes);
/* Page-locked host memory for frame buffers handed to hf_update_frame / hf_download_frame (full PCIe rate). */
int hf_host_malloc_pinned(size_t bytes, void** out_host_ptr);
int hf_host_free_pinned(void* host_ptr);

#ifdef __cplusplus
}
#endif
#endif /* HOPPERFLOW_H */
