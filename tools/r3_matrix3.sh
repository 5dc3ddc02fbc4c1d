#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3m3
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-brute-force "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "FAILED $tag"; tail -5 $OUT/$tag.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/$tag.json'));o=d['config']['delivery_overlap'];print('$tag', round(d['ms_per_step'],3), round(d['device_resident']['ms_per_step'],3), d['config']['host_frame_checked'], round(o['copy_stream_busy_ms_per_frame'],3), round(o['share_of_copy_time_beside_rendering'],3))"
}
for mode in 0 1 2; do
  for i in 1 2 3 4; do run O${mode}_$i RT_SHADE_ORDER=$mode RT_DELIVER_WGS=8 -- --steps 20 --warmup 5; done
  run O${mode}_s50 RT_SHADE_ORDER=$mode RT_DELIVER_WGS=8 -- --steps 50 --warmup 5
  run O${mode}_s100 RT_SHADE_ORDER=$mode RT_DELIVER_WGS=8 -- --steps 100 --warmup 5
  run O${mode}_s10 RT_SHADE_ORDER=$mode RT_DELIVER_WGS=8 -- --steps 10 --warmup 2
done
