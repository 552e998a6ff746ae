#!/bin/bash
# A/B two builds of liblspiv_hip.so in one box session (interleaved rounds): tools/ab_bench.sh libA.so libB.so
A=$1; B=$2; R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/pyorc_amd/liblspiv_hip.so /tmp/orig.so
for round in 1 2 3; do
  for v in A B; do
    eval src=\$$v
    cp $src $R/pyorc_amd/liblspiv_hip.so
    python $R/bench.py --steps 10 --warmup 2 --cpu-pairs 0 ${BENCH_ARGS} 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v round $round', d['value'], d['roofline']['kernel_ms_per_launch'])"
  done
done
cp /tmp/orig.so $R/pyorc_amd/liblspiv_hip.so
