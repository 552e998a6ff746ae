#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_project.py tests/test_masks.py tests/test_gpu_shard.py -m gpu -q --timeout 600 --durations=4 2>&1 | tail -12
