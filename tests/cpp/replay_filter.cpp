// tests/cpp/replay_filter.cpp -- Linux stand-in for the reference's only caller of the calculator,
// CHopperRender::DeliverToRenderer (reference HopperRender/HopperRender.cpp:907-1197), written
// against the drop-in headers exactly the way the filter uses the reference class: the two includes of
// HopperRender.cpp:24-25, `new OpticalFlowCalcSDR/HDR(...)`, the five virtuals, direct reads/writes of the
// public fields, the config.h macros.  The protocol state around the calculator (number of output frames,
// blending scalar, delta history + scene-change decision, governor) is the NATIVE one: hf_filter_* of
// include/hopperflow.h (hopperrender_amd/csrc/hf_filter.cpp).
// tests/test_cpp_adapter_gpu.py and tests/test_filter_gpu.py compare its output frames with the oracle.
//
//   replay_filter <hdr> <H> <W> <n_frames> <in_prefix> <out_prefix> <target_100ns> <radius> [threshold] [auto_adjust]
//                 [short_after] [short_playback_100ns]
// reads  <in_prefix><k>.bin   (k = 0..n_frames-1, contiguous NV12/P010)
// writes <out_prefix><m>.bin  (m = running output index) and prints one line per source and per output frame.
// radius 0 keeps the constructor's MIN_SEARCH_RADIUS; auto_adjust 1 runs autoAdjustSettings (HopperRender.cpp:951);
// from source frame `short_after` on the playback frame time is set to `short_playback_100ns` (a source period the
// GPU cannot meet: the governor must walk the radius back down).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "opticalFlowCalcSDR.h"
#include "opticalFlowCalcHDR.h"

int main(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "usage\n"); return 2; }
    const bool hdr = atoi(argv[1]) != 0;
    const int H = atoi(argv[2]), W = atoi(argv[3]), n = atoi(argv[4]);
    const std::string in = argv[5], out = argv[6];
    const long long rtTargetFrameTime = atoll(argv[7]);
    const int radius = atoi(argv[8]);
    const int threshold = argc > 9 ? atoi(argv[9]) : DEFAULT_SCENE_CHANGE_THRESHOLD;
    const bool autoAdjust = argc > 10 && atoi(argv[10]) != 0;
    const int shortAfter = argc > 11 ? atoi(argv[11]) : -1;
    const long long shortPlayback = argc > 12 ? atoll(argv[12]) : 0;
    const size_t bytes = (size_t)(hdr ? 2 : 1) * ((size_t)H * W + (size_t)(H / 2) * W);
    std::vector<unsigned char> inBuf(bytes), outBuf(bytes);
    OpticalFlowCalc* m_pofcOpticalFlowCalc = nullptr;
    hf_filter* host = nullptr;
    hf_filter_config fc{};
    fc.struct_size = sizeof(fc);
    fc.scene_change_threshold = threshold;
    fc.source_frame_time = 417083;                      // HopperRender.cpp:162
    fc.target_frame_time = rtTargetFrameTime;
    fc.frame_output_mode = HF_MODE_BLENDED_FRAME;
    fc.auto_adjust = autoAdjust;
    fc.active = 1;
    if (hf_filter_create(&fc, &host) != HF_OK) { fprintf(stderr, "hf_filter_create failed\n"); return 4; }
    int outIndex = 0;
    try {
        for (int k = 0; k < n; k++) {
            std::ifstream f(in + std::to_string(k) + ".bin", std::ios::binary);
            f.read((char*)inBuf.data(), (std::streamsize)bytes);
            if (!f) { fprintf(stderr, "cannot read frame %d\n", k); return 3; }
            if (m_pofcOpticalFlowCalc == nullptr) {  // HopperRender.cpp:907-925
                if (hdr) m_pofcOpticalFlowCalc = new OpticalFlowCalcHDR(H, W, W, W, DEFAULT_DELTA_SCALAR, DEFAULT_NEIGHBOR_SCALAR, DEFAULT_BLACK_LEVEL, DEFAULT_WHITE_LEVEL, MAX_CALC_RES);
                else m_pofcOpticalFlowCalc = new OpticalFlowCalcSDR(H, W, W, W, DEFAULT_DELTA_SCALAR, DEFAULT_NEIGHBOR_SCALAR, DEFAULT_BLACK_LEVEL, DEFAULT_WHITE_LEVEL, MAX_CALC_RES);
                if (m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius != MIN_SEARCH_RADIUS) { fprintf(stderr, "ctor radius\n"); return 6; }
                if (radius > 0) m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius = radius;  // the governor's field (HopperRender.cpp:1448,1457)
            }
            if (k == shortAfter) hf_filter_set_playback_frame_time(host, shortPlayback);
            const int m_iNumIntFrames = hf_filter_begin_source_frame(host);             // :944-948
            if (autoAdjust) {                                                            // :951 autoAdjustSettings
                int32_t r = m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius;
                hf_filter_auto_adjust(host, m_pofcOpticalFlowCalc->m_ofcCalcTime, &r);
                m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius = r;
            }
            m_pofcOpticalFlowCalc->updateFrame(inBuf.data());                            // :953
            if (m_pofcOpticalFlowCalc->m_frameCount >= 3) {                              // :955-972
                m_pofcOpticalFlowCalc->calculateOpticalFlow();
                hf_filter_push_frame_delta(host, m_pofcOpticalFlowCalc->m_frameCount, m_pofcOpticalFlowCalc->m_totalFrameDelta);
            }
            printf("src %d radius %d delta %u ofc_us %.1f\n", k, m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius,
                   m_pofcOpticalFlowCalc->m_totalFrameDelta, 1e6 * m_pofcOpticalFlowCalc->m_ofcCalcTime);
            for (int i = 0; i < m_iNumIntFrames; ++i) {
                const bool sceneChangeDetected = hf_filter_detect_scene_change(host, m_pofcOpticalFlowCalc->m_frameCount) != 0;   // :1126-1176
                const double m_dBlendingScalar = hf_filter_blending_scalar(host);
                if (m_pofcOpticalFlowCalc->m_frameCount >= 3 && !sceneChangeDetected) {  // :1179-1183
                    m_pofcOpticalFlowCalc->warpFrames((float)m_dBlendingScalar, 2);
                    printf("out %d warp %.9g delta %u\n", outIndex, (float)m_dBlendingScalar, m_pofcOpticalFlowCalc->m_totalFrameDelta);
                } else {
                    m_pofcOpticalFlowCalc->copyFrame();
                    printf("out %d copy %d\n", outIndex, sceneChangeDetected ? 1 : 0);
                }
                m_pofcOpticalFlowCalc->downloadFrame(outBuf.data());                     // :1186
                hf_filter_add_warp_duration(host, m_pofcOpticalFlowCalc->m_warpCalcTime); // :1189
                std::ofstream o(out + std::to_string(outIndex++) + ".bin", std::ios::binary);
                o.write((const char*)outBuf.data(), (std::streamsize)bytes);
                hf_filter_advance_blending_scalar(host);                                 // :1192-1197
            }
        }
        // error contract: warpFrames(t > 1) throws std::runtime_error (opticalFlowCalcSDR.cpp:143-146)
        bool threw = false;
        try { m_pofcOpticalFlowCalc->warpFrames(1.5f, 2); } catch (const std::runtime_error& e) { threw = true; }
        printf("throws_on_bad_scalar %d\n", threw ? 1 : 0);
        printf("ofc_calc_time_positive %d\n", m_pofcOpticalFlowCalc->m_ofcCalcTime > 0.0 ? 1 : 0);
        hf_filter_state st{};
        hf_filter_get_state(host, &st);
        printf("peak_scene_change_delta %u %u\n", st.peak_scene_change_delta, st.peak_scene_change_delta2);
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        delete m_pofcOpticalFlowCalc;
        hf_filter_destroy(host);
        return 5;
    }
    delete m_pofcOpticalFlowCalc;  // the filter deletes the object to force re-init (HopperRender.cpp:762-765)
    hf_filter_destroy(host);
    return 0;
}
