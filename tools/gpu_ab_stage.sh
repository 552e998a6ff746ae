#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/ab
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft32_kernel or walking_kernel or input_layouts or ragged" 2>&1 | tail -3 ) > gpurun_out/ab/pytest.log 2>&1
cp pyorc_amd/liblspiv_hip.so /tmp/base.so
for round in 1 2 3; do for v in base nostage; do
  if [ $v = base ]; then cp /tmp/base.so pyorc_amd/liblspiv_hip.so; else cp build/ab/lib_$v.so pyorc_amd/liblspiv_hip.so; fi
  echo "$v round $round: $(python tools/dtype_bench.py 2>&1 | grep 'float32  win 32' )"
done; done > gpurun_out/ab/stage.log 2>&1
cp /tmp/base.so pyorc_amd/liblspiv_hip.so
tail -2 gpurun_out/ab/pytest.log; cat gpurun_out/ab/stage.log
