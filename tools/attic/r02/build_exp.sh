#!/bin/bash
# build an experimental copy of libhopperflow.so with extra -D flags into tools/r02/exp/<name>/ (select it with HF_LIB=...)
name=$1; shift
D=tools/r02/exp/$name; mkdir -p $D
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include"
for f in hf_kernels.hip hf_flow.hip hf_capi.hip hf_filter.cpp; do /opt/rocm/bin/hipcc $F "$@" -c hopperrender_amd/csrc/$f -o $D/$f.o || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libhopperflow.so $D/*.o -Wl,-rpath,/opt/rocm/lib && rm $D/*.o && echo built $D/libhopperflow.so
