#!/bin/bash
# development probe (GPU box): alternate builds of the library on one workload; prints the isolated kernel times
# (frames rendered alone on one stream) per run
# usage: LIBS="raytracer_amd/librt_amd.so raytracer_amd/librt_x.so" bash tools/ab_probe.sh [workload] [repeats]
WL=${1:-pool_n16_4k}; REP=${2:-2}
mkdir -p gpurun_out
for i in $(seq $REP); do
for LIB in $LIBS; do
  RT_AMD_LIB=$PWD/$LIB timeout -k 10 100 python -u bench.py --workload $WL --no-cpu-baseline --no-brute-force --no-device-resident --pipelines 1 --steps 4 --warmup 3 > gpurun_out/ab_bench.json 2>gpurun_out/ab_bench.err || exit 2
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab_bench.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_frame_rank0']['isolated']
print('$LIB', 'trace %.3f shade %.3f main %.3f  frame %.3f  samples/frame %.3e' % (k['ms_trace'], k['ms_shade'], k['ms_shade_main'], d['ms_per_step'], d['config']['shadow_samples_generated_per_frame']))"
done
done
