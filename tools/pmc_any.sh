#!/bin/bash
# development probe (GPU box): arbitrary counters (one pass: at most 8 SQ_*, 4 TCC_*, ...) per kernel of the current build
#   COUNTERS="SQ_WAVES SQ_BUSY_CYCLES" bash tools/pmc_any.sh [workload]
export TMPDIR=/tmp
WL=${1:-pool_n16_4k}
rm -rf gpurun_out/pmc_any
timeout -s KILL 120 rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d gpurun_out/pmc_any -- python3 bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-brute-force --no-device-resident --pipelines 1 > gpurun_out/pmc_any.out 2>&1 || { tail -5 gpurun_out/pmc_any.out; exit 1; }
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_any/*/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][:34]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,v in acc.items():
    if 'k_' in k: print(k, 'ms', round(sum(dur[k])/len(dur[k]),3), {c:round(sum(x)/len(x)/1e6,2) for c,x in v.items()})
PY
