#!/bin/bash
# round 5, session a: the 64 x 64 ensemble kernel with half of each partial sum resident in LDS (build/ab/lib_enshalf.so =
# docs/experiments/ens64_half_lds_accumulator.patch.txt): same bits as the tree's kernel? time? HBM traffic?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5a
V=$R/build/ab/lib_enshalf.so
for thr in -1 0.2; do
  python tools/ens_hash.py 64 48 120 $thr | tail -1 | sed 's/^/tree    /'
  LSPIV_LIBRARY=$V python tools/ens_hash.py 64 48 120 $thr | tail -1 | sed 's/^/enshalf /'
done
LSPIV_LIBRARY=$V timeout 600 python -m pytest tests/test_gpu_strip_order.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "strip or ensemble" 2>&1 | tail -3
for round in 1 2; do
  python tools/ens_launch.py 64 48 1000 6 | cut -c60-140 | sed 's/^/tree-ens64 /'
  LSPIV_LIBRARY=$V python tools/ens_launch.py 64 48 1000 6 | cut -c60-140 | sed 's/^/half-ens64 /'
done
export LSPIV_LIBRARY=$V
bash tools/profile_ens.sh r05a 64 48 2>&1 | tail -30
