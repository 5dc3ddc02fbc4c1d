#!/bin/bash
# development probe (GPU box): frame times per workload and tail depth (RT_TAIL_FROM: 0 per-level launches, -2 single launch, -1 automatic)
for WL in ${WORKLOADS:-pool_n16_4k pool_4k cubes_1080p v7_1080p}; do
for T in ${TAILS:-0 -2 1 2 -1}; do
  echo "== $WL tail_from $T" >> gpurun_out/tail_probe.log
  RT_TAIL_FROM=$T timeout -k 10 100 python -u bench.py --workload $WL --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/tail_bench.json 2>gpurun_out/tail_bench.err || exit 2
  python3 -c "
import json
d=json.loads(open('gpurun_out/tail_bench.json').read().strip().splitlines()[-1])
print(d['kernel_ms_per_frame_rank0'], d['ms_per_step'])" >> gpurun_out/tail_probe.log
done
done
