#!/bin/bash
# A/B: the same workloads through the build of an older commit checked out under _old/ (scratch, not tracked)
set -o pipefail
cd _old || exit 1
export TMPDIR=/tmp
for WL in "$@"; do
  timeout -k 10 200 python3 bench.py --workload $WL --steps 10 --warmup 3 --pipelines 1 --no-cpu-baseline --no-brute-force > ../gpurun_out/old_$WL.json 2> ../gpurun_out/old_$WL.err || { tail -5 ../gpurun_out/old_$WL.err; exit 2; }
  python3 -c "import json;d=json.load(open('../gpurun_out/old_$WL.json'));k=d['kernel_ms_per_frame_rank0'];print('OLD $WL', 'frame', round(d['device_resident']['ms_per_step'],3), 'iso', {a:round(b,3) for a,b in k['isolated'].items()})"
done
