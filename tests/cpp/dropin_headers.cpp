// tests/cpp/dropin_headers.cpp -- compile-only check of the C++ drop-in surface: this translation unit includes the
// calculator headers exactly the way the reference's filter does (HopperRender.cpp:24-25; HopperRender.h includes
// nothing of the calculator) and uses what the filter uses from them WITHOUT including config.h itself (the filter gets
// it transitively through opticalFlowCalc.h:8): the config.h macros at HopperRender.cpp:180-183,1445-1457,1529-1569,
// the constructor calls at :919-922, the five virtuals and the public fields at :840,953-957,1179-1189,1331-1349,
// 1386-1389.  Built with plain g++ (no HIP, no OpenCL, no Windows headers) by tests/test_dropin_headers.py.
#include "opticalFlowCalcSDR.h"
#include "opticalFlowCalcHDR.h"

#include <cmath>
#include <deque>

#ifdef max
#error "the reference header leaks a `max` macro (opticalFlowCalc.h:14); the drop-in must not"
#endif

namespace {

struct FilterLike {
    OpticalFlowCalc* m_pofcOpticalFlowCalc = nullptr;
    unsigned int m_iSceneChangeThreshold = DEFAULT_SCENE_CHANGE_THRESHOLD;   // HopperRender.cpp:180
    unsigned int m_iBufferFrames = DEFAULT_BUFFER_FRAMES;                    // :183
    double m_dTotalWarpDuration = 0.0;
    long long m_rtCurrPlaybackFrameTime = 417083;

    void loadSettings(int* deltaScalar, int* neighborScalar, float* blackLevel, float* whiteLevel, int* maxCalcRes) {   // :1529-1569
        *deltaScalar = DEFAULT_DELTA_SCALAR;
        *neighborScalar = DEFAULT_NEIGHBOR_SCALAR;
        *blackLevel = (float)DEFAULT_BLACK_LEVEL;
        *whiteLevel = (float)DEFAULT_WHITE_LEVEL;
        *maxCalcRes = MAX_CALC_RES;
    }

    void init(bool p010, int m_iDimY, int m_iDimX, int m_iInputStride, int m_iOutputStride) {   // :907-925
        int deltaScalar, neighborScalar, customResScalar;
        float blackLevel, whiteLevel;
        loadSettings(&deltaScalar, &neighborScalar, &blackLevel, &whiteLevel, &customResScalar);
        if (p010) m_pofcOpticalFlowCalc = new OpticalFlowCalcHDR(m_iDimY, m_iDimX, m_iInputStride, m_iOutputStride, deltaScalar, neighborScalar, blackLevel, whiteLevel, customResScalar);
        else m_pofcOpticalFlowCalc = new OpticalFlowCalcSDR(m_iDimY, m_iDimX, m_iInputStride, m_iOutputStride, deltaScalar, neighborScalar, blackLevel, whiteLevel, customResScalar);
    }

    void autoAdjustSettings() {   // :1438-1463, verbatim field / macro usage
        const double dSourceFrameTimeMS = static_cast<double>(m_rtCurrPlaybackFrameTime) / 10000000.0;
        double currMaxCalcDuration = m_pofcOpticalFlowCalc->m_ofcCalcTime + m_dTotalWarpDuration;
        if ((currMaxCalcDuration * UPPER_PERF_BUFFER) > dSourceFrameTimeMS) {
            if (m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius > MIN_SEARCH_RADIUS) m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius--;
        } else if ((currMaxCalcDuration * LOWER_PERF_BUFFER) < dSourceFrameTimeMS) {
            if (m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius < MAX_SEARCH_RADIUS) m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius++;
        }
        m_dTotalWarpDuration = 0.0;
    }

    void settingsThread(int iDeltaScalar, int iNeighborScalar, int iBlackLevel, int iWhiteLevel) {   // :1385-1390
        if (m_pofcOpticalFlowCalc != nullptr) {
            m_pofcOpticalFlowCalc->m_deltaScalar = iDeltaScalar;
            m_pofcOpticalFlowCalc->m_neighborBiasScalar = iNeighborScalar;
            m_pofcOpticalFlowCalc->m_outputBlackLevel = (float)iBlackLevel;
            m_pofcOpticalFlowCalc->m_outputWhiteLevel = (float)iWhiteLevel;
        }
    }

    void statistics(double* pdOFCCalcTime, double* pdAVG, double* pdPeak, int* piLowDimX, int* piLowDimY, int* piSearchRadius) {   // :1331-1349
        *pdOFCCalcTime = 1000.0 * m_pofcOpticalFlowCalc->m_ofcCalcTime;
        *pdAVG = 1000.0 * m_pofcOpticalFlowCalc->m_ofcAvgCalcTime;
        *pdPeak = 1000.0 * m_pofcOpticalFlowCalc->m_ofcPeakCalcTime;
        *piLowDimX = m_pofcOpticalFlowCalc->m_opticalFlowFrameWidth;
        *piLowDimY = m_pofcOpticalFlowCalc->m_opticalFlowFrameHeight;
        *piSearchRadius = m_pofcOpticalFlowCalc->m_opticalFlowSearchRadius;
    }

    void deliver(unsigned char* pInBuffer, unsigned char* pOutNewBuffer, double m_dBlendingScalar, int m_iFrameOutput) {   // :953-957,1179-1189
        autoAdjustSettings();
        m_pofcOpticalFlowCalc->updateFrame(pInBuffer);
        if (m_pofcOpticalFlowCalc->m_frameCount >= 3) m_pofcOpticalFlowCalc->calculateOpticalFlow();
        unsigned int totalDelta = m_pofcOpticalFlowCalc->m_totalFrameDelta;
        (void)totalDelta;
        if (m_pofcOpticalFlowCalc->m_frameCount >= 3) m_pofcOpticalFlowCalc->warpFrames(m_dBlendingScalar, m_iFrameOutput);
        else m_pofcOpticalFlowCalc->copyFrame();
        m_pofcOpticalFlowCalc->downloadFrame(pOutNewBuffer);
        m_dTotalWarpDuration += m_pofcOpticalFlowCalc->m_warpCalcTime;
    }

    void newSegment() { if (m_pofcOpticalFlowCalc) m_pofcOpticalFlowCalc->m_frameCount = 0; }   // :840
    void reinit() { delete m_pofcOpticalFlowCalc; m_pofcOpticalFlowCalc = nullptr; }            // :762-765
};

}  // namespace

static_assert(MIN_SEARCH_RADIUS == 5 && MAX_SEARCH_RADIUS == 16 && MAX_CALC_RES == 270 && CALC_TIME_INTERVAL == 240, "config.h:4-17");
static_assert(DEFAULT_DELTA_SCALAR == 8 && DEFAULT_NEIGHBOR_SCALAR == 6 && DEFAULT_BLACK_LEVEL == 0 && DEFAULT_WHITE_LEVEL == 255, "config.h:23-26");
static_assert(DEFAULT_SCENE_CHANGE_THRESHOLD == 200 && DEFAULT_BUFFER_FRAMES == 0 && NUM_ITERATIONS == 0, "config.h:6,27-28");

int dropin_headers_anchor() {
    FilterLike f;
    (void)&FilterLike::init; (void)&FilterLike::deliver; (void)&FilterLike::settingsThread; (void)&FilterLike::statistics;
    (void)&FilterLike::newSegment; (void)&FilterLike::reinit;
    return (UPPER_PERF_BUFFER == 1.4 && LOWER_PERF_BUFFER == 1.6) ? (int)f.m_iSceneChangeThreshold : -1;
}
