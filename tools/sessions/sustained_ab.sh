#!/bin/bash
# sustained A/B of two builds: bench.py with many steps, alternating, so that clock / power management shows up
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for r in 1 2 3; do
  for v in noprio prio; do
    if [ $v = prio ]; then unset LSPIV_LIBRARY; else export LSPIV_LIBRARY=$PWD/pyorc_amd/liblspiv_hip_$v.so; fi
    timeout 200 python bench.py --gpus 1 --steps ${STEPS:-150} --warmup 5 --cpu-pairs 0 --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v r$r', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"
  done
done
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head -8
