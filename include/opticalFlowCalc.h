// include/opticalFlowCalc.h -- source-compatible replacement of the reference's calculator classes
// (reference HopperRender/opticalFlowCalc.h:24-138, opticalFlowCalcSDR.h:10-56,
// opticalFlowCalcHDR.h:10-56), implemented on the C ABI of include/hopperflow.h.
//
// What the reference's filter does with the calculator (HopperRender.cpp:907-1189) compiles unchanged
// against this header: same class names, same constructor signature, the same five virtuals, the same
// public field names and types.  Differences, all deliberate:
//   * no <CL/cl.h>, no OutputDebugStringA, no `max` macro leak (reference opticalFlowCalc.h:2-14);
//   * the cl_* handles, grids and kernels (opticalFlowCalc.h:51-86), which the filter never touches,
//     are gone; the device state lives behind the opaque hf_ctx;
//   * ownership of the public fields follows the reference.  The fields the CALLER writes -- m_deltaScalar,
//     m_neighborBiasScalar, m_outputBlackLevel, m_outputWhiteLevel (settings thread, HopperRender.cpp:1385-1390),
//     m_opticalFlowSearchRadius (governor, :1445-1458) and m_frameCount (NewSegment, :840) -- are only ever READ by this
//     class: every call hands their current values to the context first, which is when the reference's kernels see
//     them (clSetKernelArg per call, opticalFlowCalcSDR.cpp:85-86,160-161), and nothing is written back, so a value
//     poked by another thread while a blocking call is in flight survives and is used by the next call, as in the
//     reference.  updateFrame() increments m_frameCount in place (opticalFlowCalcSDR.cpp:28).  The fields the
//     CALCULATOR owns -- m_totalFrameDelta, the timings, the geometry -- are refreshed after every call;
//   * init(...) / blendFrames(t) exist as the aliases BASELINE.json's north_star names.
// Errors are std::runtime_error with the reference's "[HopperRender] ..." prefix (opticalFlowCalc.h:15-22).
#pragma once

#include <stdexcept>

#include "config.h"      // the reference reaches config.h through this header (opticalFlowCalc.h:8)
#include "hopperflow.h"

class OpticalFlowCalc {
public:
    // Video properties (opticalFlowCalc.h:27-32)
    int m_frameWidth = 0;
    int m_frameHeight = 0;
    int m_inputStride = 0;
    int m_outputStride = 0;
    float m_outputBlackLevel = 0.0f;
    float m_outputWhiteLevel = 255.0f;

    // Optical flow calculation (opticalFlowCalc.h:35-48)
    int m_opticalFlowResScalar = 0;
    int m_opticalFlowFrameWidth = 0;
    int m_opticalFlowFrameHeight = 0;
    int m_opticalFlowSearchRadius = 5;
    double m_ofcCalcTime = 0.0;
    double m_ofcAvgCalcTime = 0.0;
    double m_ofcPeakCalcTime = 0.0;
    int m_ofcCalcCount = 0;
    double m_ofcCalcTimeSum = 0.0;
    double m_warpCalcTime = 0.0;
    int m_deltaScalar = 8;
    int m_neighborBiasScalar = 6;
    unsigned int m_totalFrameDelta = 0;
    unsigned int m_frameCount = 0;

    OpticalFlowCalc() = default;
    virtual ~OpticalFlowCalc();
    OpticalFlowCalc(const OpticalFlowCalc&) = delete;
    OpticalFlowCalc& operator=(const OpticalFlowCalc&) = delete;

    virtual void updateFrame(unsigned char* inputPlanes) = 0;
    virtual void downloadFrame(unsigned char* outputPlanes) = 0;
    virtual void calculateOpticalFlow() = 0;
    virtual void warpFrames(const float blendingScalar, const int frameOutputMode) = 0;
    virtual void copyFrame() = 0;

    // north_star aliases
    void blendFrames(const float blendingScalar) { warpFrames(blendingScalar, HF_MODE_BLENDED_FRAME); }

    hf_ctx* context() const { return m_ctx; }

protected:
    // "init" = the constructor body of the reference (opticalFlowCalcSDR.cpp:206-325)
    void init(bool hdr, int frameHeight, int frameWidth, int inputStride, int outputStride, int deltaScalar,
              int neighborScalar, float blackLevel, float whiteLevel, int maxCalcRes);
    void push();            // caller-owned public fields -> context
    void pull();            // context -> calculator-owned public fields
    void check(int rc, const char* func);
    hf_ctx* m_ctx = nullptr;
};

// Shared implementation of the two reference subclasses (they differ only in the element type).
class OpticalFlowCalcImpl : public OpticalFlowCalc {
public:
    void updateFrame(unsigned char* inputPlanes) override;
    void downloadFrame(unsigned char* outputPlanes) override;
    void calculateOpticalFlow() override;
    void warpFrames(const float blendingScalar, const int frameOutputMode) override;
    void copyFrame() override;
};

class OpticalFlowCalcSDR : public OpticalFlowCalcImpl {   // NV12, 8 bit
public:
    OpticalFlowCalcSDR(const int frameHeight, const int frameWidth, const int inputStride, const int outputStride,
                       int deltaScalar, int neighborScalar, float blackLevel, float whiteLevel, int maxCalcRes);
};

class OpticalFlowCalcHDR : public OpticalFlowCalcImpl {   // P010, 16 bit
public:
    OpticalFlowCalcHDR(const int frameHeight, const int frameWidth, const int inputStride, const int outputStride,
                       int deltaScalar, int neighborScalar, float blackLevel, float whiteLevel, int maxCalcRes);
};
