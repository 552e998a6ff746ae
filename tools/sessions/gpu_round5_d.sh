#!/bin/bash
# round 5, session d: get_ffpiv writing into the run's arrays (piv_pairs out=), the budget test of borrowed ensemble chunks, project_hip
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_bench.py -m gpu -q --timeout 600 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "get_ffpiv or get_piv or chunk or device or lazy" 2>&1 | tail -3
for s in 801 802 803; do timeout 200 python tools/fuzz_modes.py $s 80 2>&1 | grep -E "FAIL|cases," | tail -2; done
python tools/hostfed_small.py 2>&1 | grep get_piv
