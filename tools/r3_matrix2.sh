#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3m2
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline --no-brute-force "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "FAILED $tag"; tail -5 $OUT/$tag.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/$tag.json'));print('$tag', round(d['ms_per_step'],3), round(d['device_resident']['ms_per_step'],3), d['config']['host_frame_checked'], d['config']['delivery_overlap'])"
}
for i in 1 2 3; do run A$i RT_SHADE_ORDER=0 RT_DELIVER=memcpy -- --steps 20 --warmup 5; done
run A_s50 RT_SHADE_ORDER=0 RT_DELIVER=memcpy -- --steps 50 --warmup 5
run A_s100 RT_SHADE_ORDER=0 RT_DELIVER=memcpy -- --steps 100 --warmup 5
run A_s10 RT_SHADE_ORDER=0 RT_DELIVER=memcpy -- --steps 10 --warmup 2
run A_p3 RT_SHADE_ORDER=0 RT_DELIVER=memcpy -- --steps 20 --warmup 5 --pipelines 3
run B_wg4 RT_SHADE_ORDER=0 RT_DELIVER_WGS=4 -- --steps 20 --warmup 5
run B_wg8 RT_SHADE_ORDER=0 RT_DELIVER_WGS=8 -- --steps 20 --warmup 5
run B_wg16 RT_SHADE_ORDER=0 RT_DELIVER_WGS=16 -- --steps 20 --warmup 5
run B_wg32 RT_SHADE_ORDER=0 RT_DELIVER_WGS=32 -- --steps 20 --warmup 5
env | grep -i -E "^(HSA|HIP|ROC|GPU|AMD)" || true
