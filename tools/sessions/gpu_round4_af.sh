#!/bin/bash
# round 4, session af: the 64 x 64 transposes through HALF the tile (lane halves exchange blocks with v_permlane32_swap, then 32 x 32
# block transposes in place; a scratch build, pyorc_amd/liblspiv_ab_halftile.so -- docs/next_round.md): correct? and what does it cost?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/pyorc_amd/liblspiv_ab_halftile.so
LSPIV_LIBRARY=$V timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -k "config3 or (full_width and 64) or (wider and 64) or other_window_size" 2>&1 | tail -2
for round in 1 2; do
  LSPIV_RESCUE=0 python tools/ab_time.py --window 64 --overlap 48 --tag tree-c3 | tail -1
  LSPIV_RESCUE=0 LSPIV_LIBRARY=$V python tools/ab_time.py --window 64 --overlap 48 --tag half-c3 | tail -1
  python tools/ens_launch.py 64 48 1000 6 | cut -c60-140 | sed 's/^/tree-ens64 /'
  LSPIV_LIBRARY=$V python tools/ens_launch.py 64 48 1000 6 | cut -c60-140 | sed 's/^/half-ens64 /'
done
