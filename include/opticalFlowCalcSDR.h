// include/opticalFlowCalcSDR.h -- forwarder: the reference's filter includes "opticalFlowCalcSDR.h" and
// "opticalFlowCalcHDR.h" (HopperRender.cpp:24-25); both classes live in opticalFlowCalc.h here
// (reference: opticalFlowCalcSDR.h:10-56 declares `class OpticalFlowCalcSDR : public OpticalFlowCalc`).
#pragma once
#include "opticalFlowCalc.h"
