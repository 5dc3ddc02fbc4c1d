#!/bin/bash
# development probe (GPU box): alternate two builds of the library on one workload; prints kernel times per run
# usage: LIB_A=... LIB_B=... [ENV_A="K=V"] [ENV_B="K=V"] bash tools/ab_probe.sh [workload] [repeats]
WL=${1:-pool_n16_4k}; REP=${2:-3}
for i in $(seq $REP); do
for V in A B; do
  if [ $V = A ]; then LIB=$LIB_A; EXTRA=$ENV_A; else LIB=$LIB_B; EXTRA=$ENV_B; fi
  env RT_AMD_LIB=$PWD/$LIB $EXTRA timeout -k 10 100 python -u bench.py --workload $WL --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/ab_bench.json 2>gpurun_out/ab_bench.err || exit 2
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab_bench.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_frame_rank0']
print('$V $LIB $EXTRA', 'trace %.3f shade %.3f frame %.3f' % (k['trace'], k['shade'], d['ms_per_step']))" >> gpurun_out/ab_probe.log
done
done
