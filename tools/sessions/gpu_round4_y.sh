#!/bin/bash
# round 4, session y: the ensemble rescue on wide grids
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "rescue_other_kernels" 2>&1 | tail -8
