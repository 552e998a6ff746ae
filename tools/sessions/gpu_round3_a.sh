#!/bin/bash
# round 3, first GPU session: rescue pass correctness on the calibration stacks + A/B against the round-2 build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r3a
{
timeout 300 python tools/calib_rescue.py --pairs 24 --tag c2_rescued 2>&1 | tail -2
timeout 300 python tools/calib_rescue.py --pairs 6 --window 64 --overlap 48 --tag c3_rescued 2>&1 | tail -2
timeout 300 python tools/calib_rescue.py --pairs 12 --dtype f32 --tag c2f32_rescued 2>&1 | tail -2
for round in 1 2 3; do
  LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_hip_r02.so timeout 120 python tools/ab_time.py --tag r02 2>&1 | tail -1
  timeout 120 python tools/ab_time.py --tag r03_rescue 2>&1 | tail -1
  LSPIV_RESCUE=0 timeout 120 python tools/ab_time.py --tag r03_norescue 2>&1 | tail -1
done
for round in 1 2; do
  LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_hip_r02.so timeout 120 python tools/ab_time.py --window 64 --overlap 48 --reps 3 --tag r02 2>&1 | tail -1
  timeout 120 python tools/ab_time.py --window 64 --overlap 48 --reps 3 --tag r03_rescue 2>&1 | tail -1
  LSPIV_RESCUE=0 timeout 120 python tools/ab_time.py --window 64 --overlap 48 --reps 3 --tag r03_norescue 2>&1 | tail -1
done
timeout 400 python bench.py --steps 5 --warmup 1 --no-extras 2>gpurun_out/r3a/bench.err | tee gpurun_out/r3a/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('bench', d['value'], {k:v for k,v in c.items() if k.startswith('parity') and k!='parity_ill_posed'})
print(json.dumps(c.get('parity_ill_posed'))[:1500])
"
} 2>&1 | tee gpurun_out/r3a/log.txt
