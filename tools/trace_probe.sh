#!/bin/bash
# development probe (GPU box): kernel trace of one rank's share of the headline frame at world size $1 (default 8)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_trace
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -- python3 tools/stripe_timing.py ${1:-8} > gpurun_out/prof_trace.log 2>&1
