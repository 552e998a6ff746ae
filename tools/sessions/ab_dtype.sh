#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/pyorc_amd/liblspiv_hip.so /tmp/orig.so
for round in 1 2; do for src in "$@"; do cp $src $R/pyorc_amd/liblspiv_hip.so; echo "$src round $round"; python $R/tools/dtype_bench.py 300 | grep "32/16"; done; done
cp /tmp/orig.so $R/pyorc_amd/liblspiv_hip.so
