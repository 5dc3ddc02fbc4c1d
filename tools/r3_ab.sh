#!/bin/bash
# A/B of library variants built by tools/build_variant.sh:  tools/r3_ab.sh "tagA tagB" workload [workload ...]
set -o pipefail
TAGS=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
for WL in "$@"; do
  for T in $TAGS; do
    LIB=raytracer_amd/librt_$T.so
    [ "$T" = base ] && LIB=raytracer_amd/librt_amd.so
    RT_AMD_LIB=$PWD/$LIB timeout -k 10 200 python3 bench.py --workload $WL --steps 10 --warmup 3 --pipelines 1 --no-cpu-baseline --no-brute-force --no-pool-4k > gpurun_out/ab/${T}_$WL.json 2> gpurun_out/ab/${T}_$WL.err || { tail -5 gpurun_out/ab/${T}_$WL.err; exit 2; }
    python3 -c "import json;d=json.load(open('gpurun_out/ab/${T}_$WL.json'));k=d['kernel_ms_per_frame_rank0'];print('$T $WL', 'frame', round(d['device_resident']['ms_per_step'],3), 'main', round(k['shade_main_kernel'],3), 'iso', {a:round(b,3) for a,b in k['isolated'].items()})"
  done
done
