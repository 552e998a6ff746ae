#!/bin/bash
# round 4, session h: suite on the new staging threads / offset guard, host-fed rates (threads sweep), ensemble rates after the epilogue reordering
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
nproc
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 -x 2>&1 | tail -6
for t in 16 8 32; do echo "threads $t"; LSPIV_STAGE_THREADS=$t python tools/hostfed_bench.py 2>&1 | tail -4; done
python tools/ens_launch.py 32 16 1000 8 | tail -1
python tools/ens_launch.py 32 16 1000 8 | tail -1
python tools/ens_launch.py 64 48 1000 5 | tail -1
