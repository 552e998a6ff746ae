#!/bin/bash
# Round 6: project_cv with both remaps in one kernel (remap_fused_kernel) against the two passes.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out/cvfused; mkdir -p $OUT
timeout 900 python -m pytest tests/test_project.py -m gpu -x -q -k "project_cv" 2>&1 | tail -15 | tee $OUT/pytest.log
for rep in 1 2; do
  LSPIV_PROJECT_DEBUG=1 python tools/rows_launch.py project_cv 30 201 2>&1 | cut -c1-150 | sed "s/^/fused /" | tee -a $OUT/ab.log
  LSPIV_PROJECT_CV_TWO_PASS=1 python tools/rows_launch.py project_cv 30 201 2>&1 | cut -c1-150 | sed "s/^/two-pass /" | tee -a $OUT/ab.log
done
