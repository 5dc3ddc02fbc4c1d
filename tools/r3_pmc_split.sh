#!/bin/bash
# counters of the classification / sampling kernels (split build) on the headline frame, for the table in DESIGN
set -o pipefail
export TMPDIR=/tmp
export RT_STREAM_PROBE=0
for PASS in "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FLOPS_FP32 SQ_WAVES SQ_BUSY_CYCLES" "mem FETCH_SIZE WRITE_SIZE"; do
  set -- $PASS; NAME=$1; shift
  OUT=gpurun_out/pmc_split_$NAME
  rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-brute-force --no-pool-4k --no-device-resident --pipelines 1 > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
  python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
f=glob.glob(out+'/*/*_counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list); seen=set()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('void ','')
    if not k.startswith('k_shade'): continue
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,c in acc.items():
    print(k, 'ms', round(sum(dur[k])/len(dur[k]),3), {n: round(sum(v)/len(v)/1e6,2) for n,v in sorted(c.items())})
PY
done
