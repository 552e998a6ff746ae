#!/bin/bash
# round 4, session v: the ensemble slot fix -- job-order tests, wide-grid parity, the ensemble tests, ensemble rates
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_strip_order.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "strip or ensemble or job_order" 2>&1 | tail -6
python tools/ens_launch.py 64 48 1000 8 | cut -c1-200
python tools/ens_launch.py 32 16 1000 12 | cut -c1-200
python tools/ens_launch.py 32 16 1000 12 f32 | cut -c1-200
