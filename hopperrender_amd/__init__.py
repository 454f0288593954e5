"""hopperrender_amd -- MI355X-native implementation of HopperRender's OpticalFlowCalc hot path.

Layout: csrc/ (HIP kernels + C ABI + C++ adapter), capi.py (ctypes binding of include/hopperflow.h),
calc.py (Python mirror of the reference's OpticalFlowCalc interface), batch.py (frame-pair sharding),
synth.py (seeded synthetic frames), build.py (in-tree native build).
"""
__version__ = "0.1.0"
