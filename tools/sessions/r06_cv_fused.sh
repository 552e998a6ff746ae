#!/bin/bash
# Round 6: project_cv with both remaps in one kernel (remap_fused_kernel / remap_fused_f32_kernel) against the two passes.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out/cvfused; mkdir -p $OUT
timeout 900 python -m pytest tests/test_project.py -m gpu -x -q -k "project_cv" 2>&1 | tail -15 | tee $OUT/pytest.log
for rep in 1 2; do
  for row in project_cv project_cv_f32; do
  python tools/rows_launch.py $row 30 201 2>&1 | grep "^project_cv" | cut -c1-120 | sed "s/^/fused /" | tee -a $OUT/ab.log
  LSPIV_PROJECT_CV_TWO_PASS=1 python tools/rows_launch.py $row 30 201 2>&1 | grep "^project_cv" | cut -c1-120 | sed "s/^/two-pass /" | tee -a $OUT/ab.log
  done
done
