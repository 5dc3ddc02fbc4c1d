#!/bin/bash
# development probe (GPU box): VALU instruction count per kernel for several builds of the library ($LIBS)
export TMPDIR=/tmp
for L in $LIBS; do
  export RT_AMD_LIB=$PWD/$L
  rm -rf gpurun_out/pmc_probe
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_probe -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-brute-force > gpurun_out/pmc_probe.out 2>&1 || exit 1
  python3 - "$L" <<'PY' >> gpurun_out/pmc_probe.log
import csv,glob,sys,collections
f=glob.glob('gpurun_out/pmc_probe/*/*counter_collection.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='SQ_INSTS_VALU': acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
print(sys.argv[1], {k[:28]:round(sum(v)/len(v)/1e6,1) for k,v in acc.items() if 'k_' in k})
PY
done
