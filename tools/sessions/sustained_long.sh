#!/bin/bash
# 20-second runs of the C2 workload per build, alternating, with power / clock samples: does a build throttle under sustained load?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/sustained
( while true; do echo "$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'sclk\|Package Power' | tr -s ' \t' ' ' | tr '\n' '|')"; sleep 1; done ) > gpurun_out/sustained/smi.log 2>&1 &
SMI=$!
for r in 1 2; do
  for v in noprio prio; do
    if [ $v = prio ]; then unset LSPIV_LIBRARY; else export LSPIV_LIBRARY=$PWD/pyorc_amd/liblspiv_hip_$v.so; fi
    echo "$(date +%s.%N) start $v r$r" >> gpurun_out/sustained/smi.log
    timeout 300 python bench.py --gpus 1 --steps ${STEPS:-3000} --warmup 5 --cpu-pairs 0 --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v r$r', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"
  done
done
kill $SMI
grep -c . gpurun_out/sustained/smi.log; awk 'NR%3==0' gpurun_out/sustained/smi.log | cut -c1-200 | head -40
