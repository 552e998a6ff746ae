#!/bin/bash
# round-2 GPU session A: parity tests, bench line, comm plumbing, C2 profile
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r2a
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r2a/pytest.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
( LSPIV_BENCH_FORCE_COMM=1 timeout 300 python bench.py --gpus 1 --steps 5 --warmup 1 --cpu-pairs 0 --no-extras ) > gpurun_out/r2a/bench_rccl1.json 2> gpurun_out/r2a/bench_rccl1.err
( LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --pairs 300 ) > gpurun_out/r2a/bench_shm2.json 2> gpurun_out/r2a/bench_shm2.err
bash tools/profile.sh r02_c2 > gpurun_out/r2a/profile_c2.log 2>&1
tail -5 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/bench.json | cut -c1-600; cat gpurun_out/r2a/bench_rccl1.json | cut -c1-300; tail -3 gpurun_out/r2a/bench_rccl1.err; cat gpurun_out/r2a/bench_shm2.json | cut -c1-300; tail -3 gpurun_out/r2a/bench_shm2.err
