#!/bin/bash
# round 3: A/B of library builds passed as arguments (name=path ...), interleaved, C2 + C3 (+ float32), then a parity subset
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2 3; do
  for kv in "$@"; do
    LSPIV_LIBRARY=${kv#*=} timeout 120 python tools/ab_time.py --tag ${kv%%=*} 2>&1 | tail -1
  done
done
for round in 1 2; do
  for kv in "$@"; do
    LSPIV_LIBRARY=${kv#*=} timeout 120 python tools/ab_time.py --window 64 --overlap 48 --reps 3 --tag ${kv%%=*} 2>&1 | tail -1
  done
done
for kv in "$@"; do
  LSPIV_LIBRARY=${kv#*=} timeout 120 python tools/ab_time.py --dtype f32 --pairs 300 --tag ${kv%%=*} 2>&1 | tail -1
done
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "fft32 or g3 or g2 or other_window or embedded_windows or walking_kernel_segments or rescue or chunks_cut" 2>&1 | tail -2
