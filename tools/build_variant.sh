#!/bin/bash
# development helper: builds raytracer_amd/librt_<tag>.so with extra -D flags for rt_shade.hip (timing experiments,
# selected at run time with RT_AMD_LIB)
TAG=$1; shift
C=raytracer_amd/csrc
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -w"
mkdir -p $C/build_$TAG
hipcc $COMMON -ffp-contract=fast-honor-pragmas "$@" -c $C/rt_shade.hip -o $C/build_$TAG/rt_shade.o || exit 1
hipcc -shared -fPIC --offload-arch=gfx950 -o raytracer_amd/librt_$TAG.so $C/build/rt_trace.o $C/build_$TAG/rt_shade.o $C/build/rt_api.o $C/build/rt_deliver.o || exit 1
echo built raytracer_amd/librt_$TAG.so
