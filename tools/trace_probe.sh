#!/bin/bash
# development probe (GPU box): kernel trace of two headline frames; per-depth k_trace durations are read from the csv
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_trace.log 2>&1
