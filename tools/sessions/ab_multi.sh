#!/bin/bash
# interleaved rounds over any number of library builds: tools/ab_multi.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/pyorc_amd/liblspiv_hip.so /tmp/orig.so
for round in 1 2 3; do
  for src in "$@"; do
    cp $src $R/pyorc_amd/liblspiv_hip.so
    python $R/bench.py --steps 10 --warmup 2 --cpu-pairs 0 ${BENCH_ARGS} 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$src round $round', d['value'], d['roofline']['kernel_ms_per_launch'])"
  done
done
cp /tmp/orig.so $R/pyorc_amd/liblspiv_hip.so
