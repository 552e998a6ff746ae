#!/bin/bash
# the committed profile sets: kernel trace (100 steps) + PMC passes for the three single-GPU BASELINE configs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile.sh r02_c2 > gpurun_out/profile_c2.log 2>&1
BENCH_ARGS="--window 64 --overlap 48" bash tools/profile.sh r02_c3 > gpurun_out/profile_c3.log 2>&1
BENCH_ARGS="--height 2160 --width 3840" bash tools/profile.sh r02_c4 > gpurun_out/profile_c4.log 2>&1
for c in c2 c3 c4; do head -2 gpurun_out/prof_r02_$c/trace_kernel_stats.csv | tail -1 | cut -c1-160; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_r02_$c/trace.log; done
( timeout 600 python bench.py ) > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-200 gpurun_out/bench_final.json
