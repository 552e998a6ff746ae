#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2 3; do
  LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_hip_r02.so timeout 120 python tools/ab_time.py --tag r02 2>&1 | tail -1
  timeout 120 python tools/ab_time.py --tag r03 2>&1 | tail -1
  LSPIV_RESCUE=0 timeout 120 python tools/ab_time.py --tag r03_norescue 2>&1 | tail -1
done
for round in 1 2; do
LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_hip_r02.so timeout 120 python tools/ab_time.py --window 64 --overlap 48 --reps 3 --tag r02 2>&1 | tail -1
timeout 120 python tools/ab_time.py --window 64 --overlap 48 --reps 3 --tag r03 2>&1 | tail -1
done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "fft32 or g3 or g2 or other_window or embedded_windows" 2>&1 | tail -2
