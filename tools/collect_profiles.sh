#!/bin/bash
# after `gpurun -- bash tools/gpu_round4_l.sh`: copy what the closing session wrote under gpurun_out/ into profiles/ (tracked)
cd "$(dirname "$0")/.."
for c in c2 c3 c4 f32 ens32 ens64; do
  for f in gpurun_out/prof_r04_$c/*_counter_collection.csv gpurun_out/prof_r04_$c/trace_kernel_stats.csv; do [ -f $f ] && cp $f profiles/r04_${c}_$(basename $f); done
  cp gpurun_out/r04_out/r04_${c}_summary.json profiles/
done
cp gpurun_out/r04_out/bench.json profiles/r04_bench.json
cp gpurun_out/r04_out/ens_rescue_cost.log profiles/r04_ens_rescue_cost.log
for f in bench_2ranks_shm bench_2ranks_shm_strong bench_torchrun_2ranks_shm bench_rccl_1rank; do cp gpurun_out/r4l/$f.json profiles/r04_$f.json; done
python - <<'PY'
import json
from pyorc_amd import _lib
print("tree kernel hash", _lib.kernel_code_hash(), "| profiles:", {c: json.load(open(f"profiles/r04_{c}_summary.json")).get("code_hash") for c in ("c2", "c3", "c4", "f32", "ens32", "ens64")})
PY
