#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r2c
( timeout 900 python -m pytest tests/test_filters.py tests/test_project.py tests/test_masks.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r2c/pytest.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
# 64 x 64 @ 75 % A/B: interleaved rounds over the variant builds (base = the shipped library)
cp pyorc_amd/liblspiv_hip.so /tmp/base.so
for round in 1 2 3; do
  for v in base sb w1 sbfft; do
    if [ $v = base ]; then cp /tmp/base.so pyorc_amd/liblspiv_hip.so; else cp build/ab/lib_$v.so pyorc_amd/liblspiv_hip.so; fi
    timeout 300 python bench.py --window 64 --overlap 48 --pairs 500 --steps 3 --warmup 1 --cpu-pairs 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v round $round', d['value'], d['roofline']['kernel_ms_per_launch'])"
  done
done > gpurun_out/r2c/ab64.log 2>&1
cp /tmp/base.so pyorc_amd/liblspiv_hip.so
BENCH_ARGS="--window 64 --overlap 48" bash tools/profile.sh r02_c3 > gpurun_out/r2c/profile_c3.log 2>&1
BENCH_ARGS="--height 2160 --width 3840" bash tools/profile.sh r02_c4 > gpurun_out/r2c/profile_c4.log 2>&1
bash tools/profile.sh r02_c2 > gpurun_out/r2c/profile_c2.log 2>&1
tail -4 gpurun_out/r2c/pytest.log; cat gpurun_out/r2c/ab64.log; cut -c1-300 gpurun_out/r2c/bench.json; tail -3 gpurun_out/r2c/bench.err
