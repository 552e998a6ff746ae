#!/bin/bash
# round 5, session e: anchor length of the walking ENSEMBLE kernels now that half of the 64 x 64 partial sum lives in LDS (round 4, all of it
# in HBM slots: 49 / 75 / 99 / 199 pairs per segment gave 29.4 / 29.9 / 28.8 / 27.9 k)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
  for w in 25 49 75 99 125; do
    LSPIV_WALK=$w python tools/ens_launch.py 64 48 1000 6 | cut -c60-140 | sed "s/^/ens64 anchor $w: /"
  done
done
for w in 25 49 75 125; do
  LSPIV_WALK=$w python tools/ens_launch.py 32 16 1000 10 | cut -c60-140 | sed "s/^/ens32 anchor $w: /"
done
