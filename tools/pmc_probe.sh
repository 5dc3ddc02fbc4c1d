#!/bin/bash
# development probe (GPU box): VALU instruction mix per kernel of the current build (or of $RT_AMD_LIB)
export TMPDIR=/tmp
WL=${1:-pool_n16_4k}
rm -rf gpurun_out/pmc_probe
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU_MUL_F32 --kernel-trace --output-format csv -d gpurun_out/pmc_probe -- python3 bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-brute-force --no-device-resident --pipelines 1 > gpurun_out/pmc_probe.out 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_probe/*/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][:34]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,v in acc.items():
    if 'k_' in k: print(k, 'ms', round(sum(dur[k])/len(dur[k]),3), {c[9:]:round(sum(x)/len(x)/1e6,1) for c,x in v.items()})
PY
