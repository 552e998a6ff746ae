#!/bin/bash
# round 5, session f: finer anchor sweep of the ensemble kernels (2000 and 1000 pairs), and the per-timestep kernels once more
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for P in 2000 1000 300; do
  for w in 25 101 125 151 175 199 249 333 499; do
    LSPIV_WALK=$w python tools/ens_launch.py 64 48 $P 4 | cut -c60-140 | sed "s/^/ens64 P=$P anchor $w: /"
  done
done
for P in 2000 1000 300; do
  for w in 25 101 125 151 175 199 249 333 499; do
    LSPIV_WALK=$w python tools/ens_launch.py 32 16 $P 8 | cut -c60-140 | sed "s/^/ens32 P=$P anchor $w: /"
  done
done
for w in 25 125 249; do
  LSPIV_WALK=$w LSPIV_RESCUE=0 python tools/ab_time.py --window 32 --overlap 16 --tag c2-anchor$w | tail -1
  LSPIV_WALK=$w LSPIV_RESCUE=0 python tools/ab_time.py --window 64 --overlap 48 --tag c3-anchor$w | tail -1
done
