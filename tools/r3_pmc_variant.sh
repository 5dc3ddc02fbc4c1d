#!/bin/bash
# VALU / SALU instruction counts of library variants (tools/build_variant.sh) on the headline frame:  tools/r3_pmc_variant.sh tag [tag ...]
set -o pipefail
export TMPDIR=/tmp
export RT_STREAM_PROBE=0
for T in "$@"; do
  LIB=$PWD/raytracer_amd/librt_$T.so
  [ "$T" = base ] && LIB=$PWD/raytracer_amd/librt_amd.so
  export RT_AMD_LIB=$LIB
  OUT=gpurun_out/pmcv_$T
  rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-brute-force --no-pool-4k --no-device-resident --pipelines 1 > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
  python3 - "$T" $OUT <<'PY'
import csv,glob,sys,collections
tag,out=sys.argv[1],sys.argv[2]
f=glob.glob(out+'/*/*_counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list); seen=set()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('void ','')
    if not k.startswith('k_shade_fast'): continue
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,c in acc.items():
    print(tag, k, 'ms', round(sum(dur[k])/len(dur[k]),3), {n: round(sum(v)/len(v)/1e6,1) for n,v in sorted(c.items())})
PY
done
