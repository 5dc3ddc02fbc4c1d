#!/bin/bash
# development probe (GPU box): kernel timeline of one rank's share of the headline frame (world given), one stream
#   bash tools/share_trace.sh 8
export TMPDIR=/tmp
W=${1:-8}
rm -rf gpurun_out/share_trace
timeout -s KILL 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/share_trace -- python3 tools/two_stream_probe.py --one-lane $W > gpurun_out/share_trace.out 2>&1 || { tail -5 gpurun_out/share_trace.out; exit 1; }
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/share_trace/*/*kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if not r['Kernel_Name'].startswith('spin')]
# last 40 launches: start offset, duration, gap to previous end
tail=rows[-36:]
t0=int(tail[0]['Start_Timestamp']); prev=None
for r in tail:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%9.1f us  dur %7.1f  gap %6.1f  %s' % ((s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3 if prev else 0, r['Kernel_Name'].split('(')[0][:40]))
    prev=e
PY
