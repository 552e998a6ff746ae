#!/bin/bash
# Round 6: anchor length of the walking kernels on MID-SIZE grids (below the 6144-window threshold that switches 25 -> 75 pairs)
cd $GRAFT_REPO_ROOT
for dt in 0 1; do
for w in default 51 75; do
  if [ $w = default ]; then python tools/size_sweep.py $dt; else LSPIV_WALK=$w python tools/size_sweep.py $dt; fi 2>&1 | grep -E "810x1440|785x875|540x960" | grep -E "P=  100|P=  200|P=  400|P= 1000" | sed "s/^/walk $w  /"
done
done
