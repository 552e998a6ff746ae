cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 60 python tools/gpu_round3_dbg2.py dev 3 2>&1 | tail -4
timeout 60 python tools/gpu_round3_dbg2.py dev 2>&1 | tail -4
timeout 60 python tools/gpu_round3_dbg2.py host 2>&1 | tail -4
timeout 100 python tools/gpu_round3_dbg.py 2>&1 | tail -12
