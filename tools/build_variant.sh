#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG=VALUE ...] -- an experimental copy of libhopperflow.so with extra compile flags in
# hopperrender_amd/lib/exp/NAME/ (git-ignored; travels to the GPU box).  Select it with HF_LIB=<path> (capi.py): the same
# library with different kernel parameters, never another implementation.  Only hf_kernels.hip / hf_flow.hip are rebuilt.
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/hopperrender_amd/lib/exp/$name; mkdir -p $D
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I $R/include"
for f in hf_kernels.hip hf_flow.hip; do /opt/rocm/bin/hipcc $F "$@" -c $R/hopperrender_amd/csrc/$f -o $D/$f.o & done; wait
python -m hopperrender_amd.build > /dev/null     # the unchanged objects (C ABI, hf_filter, hf_hostio) come from the product build
L=$R/hopperrender_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libhopperflow.so $D/hf_kernels.hip.o $D/hf_flow.hip.o \
    $L/hf_context.hip.o $L/hf_calc.hip.o $L/hf_batch.hip.o $L/hf_async_io.hip.o $L/hf_filter.cpp.o $L/hf_hostio.cpp.o -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
rm -f $D/*.o; echo built $D/libhopperflow.so
