#!/bin/bash
# Round 6: run-time radii -- the streaming kernel with a register ring (blur_stripr_kernel, default) against the LDS-ring kernel
# (LSPIV_BLUR_RING=1); the tile-per-block kernel (LSPIV_BLUR_TILE4=1) was measured by this script too and removed
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/blur_rt; mkdir -p $OUT
timeout 600 python -m pytest tests/test_filters.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
for rep in 1 2; do
  python tools/filters_bench.py 201 2>&1 | grep -E "edge 5\|9|k=11|edge 13|edge 5\|15" | cut -c1-130 | sed "s/^/strip /" | tee -a $OUT/ab.log
  LSPIV_BLUR_RING=1 python tools/filters_bench.py 201 2>&1 | grep -E "edge 5\|9|k=11|edge 13|edge 5\|15" | cut -c1-130 | sed "s/^/ring  /" | tee -a $OUT/ab.log
done
