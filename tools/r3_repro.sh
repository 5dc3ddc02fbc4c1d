#!/bin/bash
# Reproduces the driver's bench command line several times (run-to-run spread of the delivered-frame figure) and
# collects a kernel + memory-copy trace of it.  Run through gpurun from the repo root.
set -o pipefail
OUT=gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-brute-force > $OUT/drv_$i.json 2> $OUT/drv_$i.err || exit 1
  python3 -c "import json;d=json.load(open('$OUT/drv_$i.json'));print('drv',$i,d['ms_per_step'],d['device_resident']['ms_per_step'],d['kernel_ms_per_frame_rank0'])"
done
for i in 1 2; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-brute-force > $OUT/s10_$i.json 2> $OUT/s10_$i.err || exit 1
  python3 -c "import json;d=json.load(open('$OUT/s10_$i.json'));print('s10',$i,d['ms_per_step'],d['device_resident']['ms_per_step'])"
done
timeout -s KILL 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace20 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-brute-force --no-device-resident > $OUT/trace20.log 2>&1 || exit 2
tail -1 $OUT/trace20.log | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('traced',d['ms_per_step'])"
find $OUT/trace20 -name '*.csv' | xargs ls -la
