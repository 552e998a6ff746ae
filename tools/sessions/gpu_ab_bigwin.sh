#!/bin/bash
# windows above 64 px: A/B over variant builds build/ab/lib_<name>.so (two interleaved rounds): gpurun -- bash tools/gpu_ab_bigwin.sh name ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp pyorc_amd/liblspiv_hip.so /tmp/base.so
for round in 1 2; do for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so pyorc_amd/liblspiv_hip.so; else cp build/ab/lib_$v.so pyorc_amd/liblspiv_hip.so; fi
  while read -r ws ov hh ww pp; do
    timeout 300 python bench.py --window $ws --overlap $ov --height $hh --width $ww --pairs $pp --steps 3 --warmup 1 --cpu-pairs 0 --no-extras 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v round $round window $ws:', d['value'])"
  done <<'CFG'
96 48 1080 1920 200
128 64 1080 1920 200
120 60 2160 3840 100
72 36 1080 1920 200
CFG
done; done
cp /tmp/base.so pyorc_amd/liblspiv_hip.so
