#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r2b
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "above_64 or semantics or chunk or pipelined" 2>&1 | tail -30 ) > gpurun_out/r2b/pytest.log 2>&1
( timeout 300 python bench.py --window 128 --overlap 64 --height 2160 --width 3840 --pairs 50 --steps 3 --warmup 1 --cpu-pairs 0 --no-extras ) > gpurun_out/r2b/bench_128.json 2> gpurun_out/r2b/bench_128.err
( timeout 300 python bench.py --window 96 --overlap 48 --pairs 100 --steps 3 --warmup 1 --cpu-pairs 0 --no-extras ) > gpurun_out/r2b/bench_96.json 2> gpurun_out/r2b/bench_96.err
tail -12 gpurun_out/r2b/pytest.log; cut -c1-400 gpurun_out/r2b/bench_128.json; tail -3 gpurun_out/r2b/bench_128.err; cut -c1-400 gpurun_out/r2b/bench_96.json
