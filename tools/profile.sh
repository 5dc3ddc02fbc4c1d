#!/bin/bash
# Profiling passes of a bench workload on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <tag> [workload]
#   1. kernel trace + stats (per-kernel time): the default bench command, then the same with one stream
#   2..5. PMC passes, one counter group per run (rocprofv3 does not split counters over passes; at most 8 SQ_ counters)
# Output lands in gpurun_out/prof_<tag>[_<workload>]/ ; tools/summarize_profile.py condenses it into profiles/.
set -o pipefail
TAG=${1:-r02}
WL=${2:-pool_n16_4k}
OUT=gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
export TMPDIR=/tmp
export RT_STREAM_PROBE=0   # no spin kernels of the stream probe in the kernel statistics
# the default bench alternates its frames over two streams (kernels of consecutive frames overlap, so their
# durations include shared time); the counter passes and a second trace use one stream, where a launch runs alone
BENCH2="python3 bench.py --workload $WL --no-cpu-baseline --no-brute-force --no-pool-4k"
BENCH="python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-brute-force --no-pool-4k --no-device-resident --pipelines 1"
T="timeout -s KILL 240"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH2 > $OUT/trace.log 2>&1 || exit 1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_isolated -- $BENCH > $OUT/trace_isolated.log 2>&1 || exit 1
$T rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 2
$T rocprofv3 --pmc SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_class -- $BENCH > $OUT/pmc_class.log 2>&1 || exit 5
$T rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 3
$T rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 4
$T rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_f64 -- $BENCH > $OUT/pmc_f64.log 2>&1 || exit 6
echo done
