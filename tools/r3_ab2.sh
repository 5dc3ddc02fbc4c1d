#!/bin/bash
# A/B of library variants on the driver's command line (frames of two pipelines overlap):  tools/r3_ab2.sh "tagA tagB" [runs]
set -o pipefail
TAGS=$1; RUNS=${2:-2}
export TMPDIR=/tmp
mkdir -p gpurun_out/ab2
for i in $(seq $RUNS); do
  for T in $TAGS; do
    LIB=raytracer_amd/librt_$T.so
    [ "$T" = base ] && LIB=raytracer_amd/librt_amd.so
    RT_AMD_LIB=$PWD/$LIB timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-brute-force --no-pool-4k > gpurun_out/ab2/${T}_$i.json 2> gpurun_out/ab2/${T}_$i.err || { tail -5 gpurun_out/ab2/${T}_$i.err; exit 2; }
    python3 -c "import json;d=json.load(open('gpurun_out/ab2/${T}_$i.json'));k=d['kernel_ms_per_frame_rank0'];print('$T', 'delivered', round(d['ms_per_step'],3), 'resident', round(d['device_resident']['ms_per_step'],3), 'iso', {a:round(b,3) for a,b in k['isolated'].items()})"
  done
done
