#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r2d
( timeout 900 python -m pytest tests/test_project.py tests/test_filters.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r2d/pytest.log 2>&1
( timeout 300 python bench.py --window 64 --overlap 48 --pairs 1000 --steps 3 --warmup 1 --cpu-pairs 0 --no-extras ) > gpurun_out/r2d/bench_c3.json 2> gpurun_out/r2d/bench_c3.err
bash tools/profile.sh r02_c2 > gpurun_out/r2d/profile_c2.log 2>&1
BENCH_ARGS="--window 64 --overlap 48" bash tools/profile.sh r02_c3 > gpurun_out/r2d/profile_c3.log 2>&1
tail -5 gpurun_out/r2d/pytest.log; cut -c1-250 gpurun_out/r2d/bench_c3.json; head -2 gpurun_out/prof_r02_c2/trace_kernel_stats.csv; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_r02_c2/trace.log
