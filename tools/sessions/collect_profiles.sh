#!/bin/bash
# after a closing session (`gpurun -- bash tools/gpu_round5_z.sh`): copy what it wrote under gpurun_out/ into profiles/ (tracked)
#   usage: tools/collect_profiles.sh [tag (default r05)] [session directory under gpurun_out/ with the N > 1 plumbing lines (default r5z)]
cd "$(dirname "$0")/.."
TAG=${1:-r05}; SES=${2:-r5z}
for c in c2 c3 c4 f32 ens32 ens64; do
  for f in gpurun_out/prof_${TAG}_$c/*_counter_collection.csv gpurun_out/prof_${TAG}_$c/trace_kernel_stats.csv; do [ -f $f ] && cp $f profiles/${TAG}_${c}_$(basename $f); done
  cp gpurun_out/${TAG}_out/${TAG}_${c}_summary.json profiles/
done
cp gpurun_out/${TAG}_out/bench.json profiles/${TAG}_bench.json
cp gpurun_out/${TAG}_out/ens_rescue_cost.log profiles/${TAG}_ens_rescue_cost.log
for f in bench_2ranks_shm bench_2ranks_shm_strong bench_torchrun_2ranks_shm bench_rccl_1rank; do [ -f gpurun_out/$SES/$f.json ] && cp gpurun_out/$SES/$f.json profiles/${TAG}_$f.json; done
python - "$TAG" <<'PY'
import json, sys
from pyorc_amd import _lib
t = sys.argv[1]
print("tree kernel hash", _lib.kernel_code_hash(), "| profiles:", {c: json.load(open(f"profiles/{t}_{c}_summary.json")).get("code_hash") for c in ("c2", "c3", "c4", "f32", "ens32", "ens64")})
PY
