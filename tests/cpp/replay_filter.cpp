// tests/cpp/replay_filter.cpp -- Linux stand-in for the reference's only caller of the calculator,
// CHopperRender::DeliverToRenderer (reference HopperRender/HopperRender.cpp:907-1197), written
// against include/opticalFlowCalc.h exactly the way the filter uses the reference class:
// `new OpticalFlowCalcSDR/HDR(...)`, the five virtuals, direct reads/writes of the public fields.
// It proves the C++ surface is a drop-in: tests/test_cpp_adapter_gpu.py compares its output frames
// with the Python mirror and the oracle.
//
//   replay_filter <hdr> <H> <W> <n_frames> <in_prefix> <out_prefix> <target_100ns> <radius>
// reads  <in_prefix><k>.bin   (k = 0..n_frames-1, contiguous NV12/P010)
// writes <out_prefix><m>.bin  (m = running output index) and prints one line per output frame.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "opticalFlowCalc.h"

int main(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "usage\n"); return 2; }
    const bool hdr = atoi(argv[1]) != 0;
    const int H = atoi(argv[2]), W = atoi(argv[3]), n = atoi(argv[4]);
    const std::string in = argv[5], out = argv[6];
    const long long rtTargetFrameTime = atoll(argv[7]);
    const int radius = atoi(argv[8]);
    const long long rtSourceFrameTime = 417083, rtCurrPlaybackFrameTime = rtSourceFrameTime;  // HopperRender.cpp:162
    const size_t bytes = (size_t)(hdr ? 2 : 1) * ((size_t)H * W + (size_t)(H / 2) * W);
    std::vector<unsigned char> inBuf(bytes), outBuf(bytes);
    OpticalFlowCalc* m_pofcOpticalFlowCalc = nullptr;
    double m_dBlendingScalar = 0.0;
    int outIndex = 0;
    try {
        for (int k = 0; k < n; k++) {
            std::ifstream f(in + std::to_string(k) + ".bin", std::ios::binary);
            f.read((char*)inBuf.data(), (std::streamsize)bytes);
            if (!f) { fprintf(stderr, "cannot read frame %d\n", k); return 3; }
            if (m_pofcOpticalFlowCalc == nullptr) {  // HopperRender.cpp:907-925
                if (hdr) m_pofcOpticalFlowCalc = new OpticalFlowCalcHDR(H, W, W, W, 8, 6, 0.0f, 255.0f, 270);
                else m_pofcOpticalFlowCalc = new OpticalFlowCalcSDR(H, W, W, W, 8, 6, 0.0f, 255.0f, 270);
                m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius = radius;  // the governor's field (HopperRender.cpp:1448,1457)
            }
            // HopperRender.cpp:944-948
            const int m_iNumIntFrames = (int)std::fmax(std::ceil((1.0 - m_dBlendingScalar) / ((double)rtTargetFrameTime / (double)rtCurrPlaybackFrameTime)), 1.0);
            m_pofcOpticalFlowCalc->updateFrame(inBuf.data());                        // :953
            if (m_pofcOpticalFlowCalc->m_frameCount >= 3) m_pofcOpticalFlowCalc->calculateOpticalFlow();  // :955-957
            for (int i = 0; i < m_iNumIntFrames; ++i) {
                if (m_pofcOpticalFlowCalc->m_frameCount >= 3) {                      // :1179-1183
                    m_pofcOpticalFlowCalc->warpFrames((float)m_dBlendingScalar, 2);
                    printf("out %d warp %.9g delta %u\n", outIndex, (float)m_dBlendingScalar, m_pofcOpticalFlowCalc->m_totalFrameDelta);
                } else {
                    m_pofcOpticalFlowCalc->copyFrame();
                    printf("out %d copy\n", outIndex);
                }
                m_pofcOpticalFlowCalc->downloadFrame(outBuf.data());                 // :1186
                std::ofstream o(out + std::to_string(outIndex++) + ".bin", std::ios::binary);
                o.write((const char*)outBuf.data(), (std::streamsize)bytes);
                m_dBlendingScalar += (double)rtTargetFrameTime / (double)rtCurrPlaybackFrameTime;  // :1192-1197
                if (m_dBlendingScalar >= 1.0) m_dBlendingScalar -= 1.0;
            }
        }
        // error contract: warpFrames(t > 1) throws std::runtime_error (opticalFlowCalcSDR.cpp:143-146)
        bool threw = false;
        try { m_pofcOpticalFlowCalc->warpFrames(1.5f, 2); } catch (const std::runtime_error& e) { threw = true; }
        printf("throws_on_bad_scalar %d\n", threw ? 1 : 0);
        printf("ofc_calc_time_positive %d\n", m_pofcOpticalFlowCalc->m_ofcCalcTime > 0.0 ? 1 : 0);
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        delete m_pofcOpticalFlowCalc;
        return 5;
    }
    delete m_pofcOpticalFlowCalc;  // the filter deletes the object to force re-init (HopperRender.cpp:762-765)
    return 0;
}
