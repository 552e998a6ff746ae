#!/bin/bash
# round 3: how the parity error over all 7.85 M windows and the rescue counts move with the flag model's kappa
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r3e
for k in 300 400 450 500; do
  LSPIV_RESCUE_KAPPA=$k timeout 300 python bench.py --steps 5 --warmup 1 --no-extras 2>gpurun_out/r3e/bench_$k.err | tee gpurun_out/r3e/bench_$k.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('kappa $k', 'pairs/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms_per_launch'], {k:v for k,v in c.items() if k.startswith('parity') and not isinstance(v, dict)})
"
  LSPIV_RESCUE_KAPPA=$k timeout 100 python tools/ab_time.py --tag kappa$k | tail -1
done
