#!/bin/bash
# development helper: builds raytracer_amd/librt_<tag>.so with extra -D flags for EVERY source file (experiments that
# change a constant shared by the kernels and the host code); selected at run time with RT_AMD_LIB
TAG=$1; shift
C=raytracer_amd/csrc
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -w"
mkdir -p $C/build_$TAG
hipcc $COMMON -ffp-contract=off "$@" -c $C/rt_trace.hip -o $C/build_$TAG/rt_trace.o &
hipcc $COMMON -ffp-contract=fast-honor-pragmas "$@" -c $C/rt_shade.hip -o $C/build_$TAG/rt_shade.o &
hipcc $COMMON -ffp-contract=off "$@" -c $C/rt_api.hip -o $C/build_$TAG/rt_api.o &
hipcc $COMMON "$@" -c $C/rt_deliver.hip -o $C/build_$TAG/rt_deliver.o &
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o raytracer_amd/librt_$TAG.so $C/build_$TAG/rt_trace.o $C/build_$TAG/rt_shade.o $C/build_$TAG/rt_api.o $C/build_$TAG/rt_deliver.o || exit 1
echo built raytracer_amd/librt_$TAG.so
