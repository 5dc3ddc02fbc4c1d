#!/bin/bash
# The measurements DESIGN quotes, in one call on the GPU box: profiles of every workload, one rank's share of the headline
# frame per world size, the parity report, and the driver's bench command line.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out/final
bash tools/profile_all.sh $TAG > gpurun_out/final/profile_all.log 2>&1 || { tail -5 gpurun_out/final/profile_all.log; exit 1; }
timeout -k 10 400 env GPU_MAX_HW_QUEUES=8 python3 tools/two_stream_probe.py 1 2 4 8 > gpurun_out/final/share_scaling.txt 2>&1 || { tail -5 gpurun_out/final/share_scaling.txt; exit 2; }
timeout -k 10 600 python3 tools/golden_report.py gpurun_out/final/parity.json > gpurun_out/final/parity.log 2>&1 || { tail -5 gpurun_out/final/parity.log; exit 3; }
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_$i.json 2> gpurun_out/final/bench_$i.err || { tail -5 gpurun_out/final/bench_$i.err; exit 4; }
done
echo final done
