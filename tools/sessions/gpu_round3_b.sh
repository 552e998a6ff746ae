#!/bin/bash
# round 3: cost of the rescue pass, phase by phase, against the round-2 build (interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r3b
{
for cfg in "--window 32 --overlap 16" "--window 64 --overlap 48 --reps 3"; do
for round in 1 2; do
  LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_hip_r02.so timeout 120 python tools/ab_time.py $cfg --tag r02 2>&1 | tail -1
  timeout 120 python tools/ab_time.py $cfg --tag r03_rescue 2>&1 | tail -1
  LSPIV_RESCUE=0 timeout 120 python tools/ab_time.py $cfg --tag r03_norescue 2>&1 | tail -1
  LSPIV_RESCUE_KAPPA=0 timeout 120 python tools/ab_time.py $cfg --tag r03_amb_only 2>&1 | tail -1
  LSPIV_RESCUE_TAU=0 timeout 120 python tools/ab_time.py $cfg --tag r03_fit_only 2>&1 | tail -1
done
done
timeout 400 python bench.py --steps 5 --warmup 1 --no-extras 2>gpurun_out/r3b/bench.err | tee gpurun_out/r3b/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('bench', d['value'], {k:v for k,v in c.items() if k.startswith('parity') and not isinstance(v, dict)})
"
} 2>&1 | tee gpurun_out/r3b/log.txt
