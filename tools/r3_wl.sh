#!/bin/bash
# Frames of the given workloads rendered alone on one stream (kernel times per frame) -- run through gpurun:
#   tools/r3_wl.sh tag workload [workload ...]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/wl_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for WL in "$@"; do
  timeout -k 10 200 python3 bench.py --workload $WL --steps 10 --warmup 3 --pipelines 1 --no-cpu-baseline --no-brute-force --no-pool-4k > $OUT/$WL.json 2> $OUT/$WL.err || { tail -5 $OUT/$WL.err; exit 2; }
  python3 -c "import json;d=json.load(open('$OUT/$WL.json'));k=d['kernel_ms_per_frame_rank0'];print('$WL', 'frame', round(d['device_resident']['ms_per_step'],3), 'delivered', round(d['ms_per_step'],3), 'trace', round(k['trace'],3), 'shade', round(k['shade'],3), 'main', round(k['shade_main_kernel'],3), 'iso', {a:round(b,3) for a,b in k['isolated'].items()})"
done
