#!/bin/bash
# round 4, session ad: the A/B switches of the rows around the hot path still give the oracle's bits (non-default kernels kept for comparison)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for e in LSPIV_NORM_FRAME_MAJOR=1 LSPIV_BLUR_BLOCK=1 LSPIV_PROJECT_ONE_CELL=1 LSPIV_PROJECT_FPT=1 LSPIV_PROJECT_FPT=4 LSPIV_PROJECT_GX=0 LSPIV_STAGE_THREADS=1 LSPIV_STAGE_THREADS=7; do
  echo "$e: $(env $e timeout 300 python tools/fuzz_rows.py 41 120 2>&1 | tail -1)"
done
for e in LSPIV_NORM_FRAME_MAJOR=1 LSPIV_BLUR_BLOCK=1 LSPIV_PROJECT_ONE_CELL=1; do
  echo "$e tests: $(env $e timeout 600 python -m pytest tests/test_filters.py tests/test_project.py -m gpu -q 2>&1 | tail -1)"
done
