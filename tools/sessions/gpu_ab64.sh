#!/bin/bash
# 64 x 64 @ 75 % (BASELINE configs[2]) A/B over variant builds of the library: build/ab/lib_<name>.so, interleaved rounds,
# base = the shipped library.  usage: gpurun -- bash tools/gpu_ab64.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/ab64
cp pyorc_amd/liblspiv_hip.so /tmp/base.so
for round in 1 2 3; do
  for v in base "$@"; do
    if [ $v = base ]; then cp /tmp/base.so pyorc_amd/liblspiv_hip.so; else cp build/ab/lib_$v.so pyorc_amd/liblspiv_hip.so; fi
    timeout 300 python bench.py --window ${AB_WINDOW:-64} --overlap ${AB_OVERLAP:-48} --pairs ${AB_PAIRS:-500} --steps 3 --warmup 1 --cpu-pairs 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v round $round', d['value'], d['roofline']['kernel_ms_per_launch'])"
  done
done > gpurun_out/ab64/ab.log 2>&1
cp /tmp/base.so pyorc_amd/liblspiv_hip.so
cat gpurun_out/ab64/ab.log
