#!/bin/bash
# round 4, session t: ensemble kernels over the segment length (LSPIV_WALK = pairs per segment; default anchor 25)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
  for w in 1 49 75 99 199; do
    LSPIV_WALK=$w python tools/ens_launch.py 64 48 1000 8 | cut -c1-160 | sed "s/^/walk $w /"
    LSPIV_WALK=$w python tools/ens_launch.py 32 16 1000 12 | cut -c1-160 | sed "s/^/walk $w /"
  done
done
