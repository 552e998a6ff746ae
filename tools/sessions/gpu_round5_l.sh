#!/bin/bash
# round 5, session l: an anchor that divides 1000 (rank blocks of BASELINE configs[4]): 100 against 75 and 125, two rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
for w in 75 100 125; do
  export LSPIV_WALK=$w
  for P in 1000 1040 300; do
    LSPIV_RESCUE=0 python tools/ab_time.py --window 32 --overlap 16 --pairs $P --tag "c2 P=$P anchor $w" | tail -1
    LSPIV_RESCUE=0 python tools/ab_time.py --window 64 --overlap 48 --pairs $P --tag "c3 P=$P anchor $w" | tail -1
  done
  python tools/ens_launch.py 64 48 1000 4 | cut -c88-140 | sed "s/^/ens64 anchor $w: /"
  python tools/ens_launch.py 32 16 1000 6 | cut -c88-140 | sed "s/^/ens32 anchor $w: /"
done
done
