#!/bin/bash
# per-dtype A/B (tools/dtype_bench.py) over variant builds build/ab/lib_<name>.so, interleaved: gpurun -- bash tools/gpu_ab_dtype.sh name ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/abdtype
cp pyorc_amd/liblspiv_hip.so /tmp/base.so
for round in 1 2 3; do
  for v in base "$@"; do
    if [ $v = base ]; then cp /tmp/base.so pyorc_amd/liblspiv_hip.so; else cp build/ab/lib_$v.so pyorc_amd/liblspiv_hip.so; fi
    echo "== $v round $round"; timeout 300 python tools/dtype_bench.py 200 2>&1 | grep -v uint8
  done
done > gpurun_out/abdtype/ab.log 2>&1
cp /tmp/base.so pyorc_amd/liblspiv_hip.so
cat gpurun_out/abdtype/ab.log
