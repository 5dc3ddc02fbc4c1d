#!/bin/bash
# A/B matrix of the frame delivery at the driver's bench arguments: runtime copy against the row kernel, main
# shading launches ordered or not.  Run through gpurun from the repo root.
set -o pipefail
OUT=gpurun_out/r3m
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # tag, env..., -- args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-brute-force "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "FAILED $tag"; tail -5 $OUT/$tag.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/$tag.json'));print('$tag', round(d['ms_per_step'],3), round(d['device_resident']['ms_per_step'],3), d['config']['host_frame_checked'], d['config']['delivery_overlap'])"
}
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "frame_stream or pipelines or enqueue" --timeout 120 --timeout-method thread > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
run A_ring_only RT_SHADE_ORDER=0 RT_DELIVER=memcpy -- --steps 20 --warmup 5
run B_kernel RT_SHADE_ORDER=0 -- --steps 20 --warmup 5
run C_order RT_SHADE_ORDER=1 RT_DELIVER=memcpy -- --steps 20 --warmup 5
run D_both RT_SHADE_ORDER=1 -- --steps 20 --warmup 5
run D_both2 RT_SHADE_ORDER=1 -- --steps 20 --warmup 5
run D_s10 RT_SHADE_ORDER=1 -- --steps 10 --warmup 2
run D_s50 RT_SHADE_ORDER=1 -- --steps 50 --warmup 5
run D_wg16 RT_SHADE_ORDER=1 RT_DELIVER_WGS=16 -- --steps 20 --warmup 5
run D_wg256 RT_SHADE_ORDER=1 RT_DELIVER_WGS=256 -- --steps 20 --warmup 5
run D_p3 RT_SHADE_ORDER=1 -- --steps 20 --warmup 5 --pipelines 3
run D_p1 RT_SHADE_ORDER=1 -- --steps 20 --warmup 5 --pipelines 1
