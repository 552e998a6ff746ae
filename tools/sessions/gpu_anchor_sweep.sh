#!/bin/bash
# anchor length of the walking kernels (LSPIV_WALK=n) against throughput on the BASELINE configs: gpurun -- bash tools/gpu_anchor_sweep.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "32 16 1080 1920 1000" "64 48 1080 1920 1000" "32 16 2160 3840 500"; do set -- $cfg
  for a in 25 33 51 75 101 125; do
    LSPIV_WALK=$a timeout 300 python bench.py --window $1 --overlap $2 --height $3 --width $4 --pairs $5 --steps 5 --warmup 2 --cpu-pairs 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('window $1 frame $3x$4 pairs $5 anchor $a:', d['value'], d['roofline']['kernel_ms_per_launch'])"
  done
done
