#!/bin/bash
# Round 6: normalize in one pass over the frames (norm_onepass_kernel) against the two passes; every step under its own timeout
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/norm1; mkdir -p $OUT
timeout 180 python -m pytest tests/test_filters.py -m gpu -x -q -k "normalize" 2>&1 | tail -4 | tee $OUT/pytest.log
for rep in 1 2; do
  LSPIV_NORM_ONE_PASS=1 timeout 120 python tools/rows_launch.py normalize 30 201 2>&1 | grep "normalize:" | cut -c1-100 | sed "s/^/one-pass /" | tee -a $OUT/ab.log
  timeout 120 python tools/rows_launch.py normalize 30 201 2>&1 | grep "normalize:" | cut -c1-100 | sed "s/^/two-pass /" | tee -a $OUT/ab.log
done
