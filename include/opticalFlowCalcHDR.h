// include/opticalFlowCalcHDR.h -- forwarder, see opticalFlowCalcSDR.h (reference: opticalFlowCalcHDR.h:10-56,
// included at HopperRender.cpp:25).
#pragma once
#include "opticalFlowCalc.h"
