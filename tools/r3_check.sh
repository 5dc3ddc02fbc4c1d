#!/bin/bash
# GPU suite, then the driver's bench command line (run through gpurun from the repo root):  tools/r3_check.sh [runs]
set -o pipefail
OUT=gpurun_out/r3c
RUNS=${1:-2}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
for i in $(seq 1 $RUNS); do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv_$i.json 2> $OUT/drv_$i.err || { tail -5 $OUT/drv_$i.err; exit 2; }
  python3 -c "import json;d=json.load(open('$OUT/drv_$i.json'));print('drv',$i,round(d['ms_per_step'],3),round(d['device_resident']['ms_per_step'],3),d['kernel_ms_per_frame_rank0']['isolated'],d['config']['host_frame_checked'],d['config']['shadow_samples_generated_per_frame'])"
done
