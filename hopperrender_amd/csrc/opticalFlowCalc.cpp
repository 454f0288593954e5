// hopperrender_amd/csrc/opticalFlowCalc.cpp -- the C++ adapter of include/opticalFlowCalc.h: plain
// C++17, no HIP headers; everything goes through the C ABI (include/hopperflow.h).
#include "opticalFlowCalc.h"

#include <cstdio>
#include <string>

void OpticalFlowCalc::check(int rc, const char* func) {
    if (rc == HF_OK) return;
    std::string msg = hf_last_error(m_ctx);
    if (msg.empty()) msg = std::string("[HopperRender] Error in function ") + func + "\n";
    fputs((msg + "\n").c_str(), stderr);   // reference: CHECK_ERROR prints, then throws (opticalFlowCalc.h:15-22)
    throw std::runtime_error(msg);
}

void OpticalFlowCalc::init(bool hdr, int frameHeight, int frameWidth, int inputStride, int outputStride, int deltaScalar,
                           int neighborScalar, float blackLevel, float whiteLevel, int maxCalcRes) {
    if (m_ctx) { hf_destroy(m_ctx); m_ctx = nullptr; }
    hf_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.is_hdr = hdr ? 1 : 0;
    cfg.frame_height = frameHeight;
    cfg.frame_width = frameWidth;
    cfg.input_stride = inputStride;
    cfg.output_stride = outputStride;
    cfg.delta_scalar = deltaScalar;
    cfg.neighbor_scalar = neighborScalar;
    cfg.black_level = blackLevel;
    cfg.white_level = whiteLevel;
    cfg.max_calc_res = maxCalcRes;
    cfg.device_index = -1;   // detectDevices: the first suitable device (opticalFlowCalc.cpp:67-93), not a fixed ordinal
    const int rc = hf_create(&cfg, &m_ctx);
    if (rc != HF_OK) {
        std::string msg = hf_last_error(nullptr);
        if (msg.empty()) msg = "[HopperRender] Error in function OpticalFlowCalc\n";
        fputs((msg + "\n").c_str(), stderr);
        throw std::runtime_error(msg);
    }
    m_deltaScalar = deltaScalar;                  // opticalFlowCalcSDR.cpp:214-216,223-224
    m_neighborBiasScalar = neighborScalar;
    m_outputBlackLevel = blackLevel;
    m_outputWhiteLevel = whiteLevel;
    m_opticalFlowSearchRadius = MIN_SEARCH_RADIUS;
    m_frameCount = 0;
    pull();
    hf_stats st{};
    hf_get_stats(m_ctx, &st);
    printf("[HopperRender] Using HIP device %d and %llu MB of VRAM\n", hf_get_device(m_ctx),   // opticalFlowCalc.cpp:90
           (unsigned long long)((3 * st.input_frame_bytes + st.output_frame_bytes) / 1024 / 1024));
}

OpticalFlowCalc::~OpticalFlowCalc() {
    if (m_ctx) hf_destroy(m_ctx);   // opticalFlowCalcSDR.cpp:185-204
    m_ctx = nullptr;
}

void OpticalFlowCalc::push() {
    hf_params p{};
    p.delta_scalar = m_deltaScalar;
    p.neighbor_scalar = m_neighborBiasScalar;
    p.black_level = m_outputBlackLevel;
    p.white_level = m_outputWhiteLevel;
    p.search_radius = m_opticalFlowSearchRadius;
    p.frame_count = m_frameCount;
    check(hf_set_params(m_ctx, &p), "push");
}

// Only what the calculator owns (include/opticalFlowCalc.h): results, timings, geometry.  The caller-owned inputs are
// never written back -- the reference's settings thread pokes them while the streaming thread is inside a blocking call
// (HopperRender.cpp:1385-1390) and that write must survive the call.
void OpticalFlowCalc::pull() {
    hf_stats s{};
    check(hf_get_stats(m_ctx, &s), "pull");
    m_totalFrameDelta = s.total_frame_delta;
    m_ofcCalcTime = s.ofc_calc_time;
    m_ofcAvgCalcTime = s.ofc_avg_calc_time;
    m_ofcPeakCalcTime = s.ofc_peak_calc_time;
    m_warpCalcTime = s.warp_calc_time;
    m_opticalFlowResScalar = s.res_scalar;
    m_opticalFlowFrameWidth = s.low_width;
    m_opticalFlowFrameHeight = s.low_height;
    m_frameWidth = s.frame_width;
    m_frameHeight = s.frame_height;
    m_inputStride = s.input_stride;
    m_outputStride = s.output_stride;
}

void OpticalFlowCalcImpl::updateFrame(unsigned char* inputPlanes) {
    push();
    check(hf_update_frame(m_ctx, inputPlanes), __func__);
    m_frameCount++;   // opticalFlowCalcSDR.cpp:28 -- on the field itself: NewSegment's reset (HopperRender.cpp:840) is not overwritten
    pull();
}

void OpticalFlowCalcImpl::downloadFrame(unsigned char* outputPlanes) {
    push();
    check(hf_download_frame(m_ctx, outputPlanes), __func__);
    pull();
}

void OpticalFlowCalcImpl::calculateOpticalFlow() {
    push();
    check(hf_calculate_optical_flow(m_ctx), __func__);
    pull();
}

void OpticalFlowCalcImpl::warpFrames(const float blendingScalar, const int frameOutputMode) {
    if (blendingScalar > 1.0f) printf("Error: Blending scalar is greater than 1.0\n");   // opticalFlowCalcSDR.cpp:144
    push();
    check(hf_warp_frames(m_ctx, blendingScalar, frameOutputMode), __func__);
    pull();
}

void OpticalFlowCalcImpl::copyFrame() {
    push();
    check(hf_copy_frame(m_ctx), __func__);
    pull();
}

OpticalFlowCalcSDR::OpticalFlowCalcSDR(const int frameHeight, const int frameWidth, const int inputStride,
                                       const int outputStride, int deltaScalar, int neighborScalar, float blackLevel,
                                       float whiteLevel, int maxCalcRes) {
    init(false, frameHeight, frameWidth, inputStride, outputStride, deltaScalar, neighborScalar, blackLevel, whiteLevel, maxCalcRes);
}

OpticalFlowCalcHDR::OpticalFlowCalcHDR(const int frameHeight, const int frameWidth, const int inputStride,
                                       const int outputStride, int deltaScalar, int neighborScalar, float blackLevel,
                                       float whiteLevel, int maxCalcRes) {
    init(true, frameHeight, frameWidth, inputStride, outputStride, deltaScalar, neighborScalar, blackLevel, whiteLevel, maxCalcRes);
}
