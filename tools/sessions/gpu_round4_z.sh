#!/bin/bash
# round 4, session z: sensitivity of the float32 kernel to the number of distinct lines per load -- a build whose two jobs of a wave read
# the SAME window (wrong results, timing only; pyorc_amd/liblspiv_ab_samewin.so) against the tree's build, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2 3; do
  LSPIV_RESCUE=0 python tools/ab_time.py --dtype f32 --tag tree | tail -1
  LSPIV_RESCUE=0 LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_ab_samewin.so python tools/ab_time.py --dtype f32 --tag samewin | tail -1
  LSPIV_RESCUE=0 python tools/ab_time.py --dtype u8 --tag tree-u8 | tail -1
  LSPIV_RESCUE=0 LSPIV_LIBRARY=$R/pyorc_amd/liblspiv_ab_samewin.so python tools/ab_time.py --dtype u8 --tag samewin-u8 | tail -1
done
