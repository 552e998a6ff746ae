#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -3
for round in 1 2; do
  LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_hip_r02.so timeout 120 python tools/ab_time.py --tag r02 2>&1 | tail -1
  timeout 120 python tools/ab_time.py --tag r03 2>&1 | tail -1
  LSPIV_RESCUE=0 timeout 120 python tools/ab_time.py --tag r03_norescue 2>&1 | tail -1
done
LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_hip_r02.so timeout 120 python tools/ab_time.py --window 64 --overlap 48 --reps 3 --tag r02 2>&1 | tail -1
timeout 120 python tools/ab_time.py --window 64 --overlap 48 --reps 3 --tag r03 2>&1 | tail -1
LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_hip_r02.so timeout 120 python tools/ab_time.py --dtype f32 --pairs 300 --tag r02_f32 2>&1 | tail -1
timeout 120 python tools/ab_time.py --dtype f32 --pairs 300 --tag r03_f32 2>&1 | tail -1
bash tools/gpu_round3_c.sh 2>&1 | grep -v "synth_\|rocclr\|simple_timer"
timeout 500 python bench.py 2>gpurun_out/r3f/bench.err | tee gpurun_out/r3f/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline'].get('launch_ms_with_rescue_kernels'), d['config'].get('rescue'))
print({k:v for k,v in c.items() if k.startswith('parity') and not isinstance(v, dict)})
for o in d['config'].get('other_configs', []): print(o['workload'][:40], o['pairs_per_s'], o['launch_ms'], o['kernel_ms'], o['rescued_windows_per_launch'])
print(d['config'].get('host_fed_pairs_per_s')); print(d['config'].get('camera_to_velocity_pairs_per_s'))
"
