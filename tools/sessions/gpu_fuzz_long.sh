#!/bin/bash
# long randomised differential run: gpurun -- bash tools/gpu_fuzz_long.sh <first seed> <n seeds> <cases per seed>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/fuzz
for s in $(seq $1 $(($1 + $2 - 1))); do
  FUZZ_DUMP=gpurun_out/fuzz/dump_$s timeout 900 python tools/fuzz_parity.py $s $3 > gpurun_out/fuzz/seed_$s.log 2>&1
  echo "seed $s: $(tail -1 gpurun_out/fuzz/seed_$s.log)"; grep FAIL gpurun_out/fuzz/seed_$s.log | head -5
done
