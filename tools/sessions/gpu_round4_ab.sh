#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -k "beyond" --durations=3 2>&1 | tail -8
