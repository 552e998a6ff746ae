#!/bin/bash
# round 3: long fuzz with all windows gated + the committed profile sets of the three single-GPU BASELINE configs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/gpu_fuzz_long.sh 300 4 120 2>&1 | tail -20
bash tools/profile.sh r03_c2 > gpurun_out/profile_c2.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r03_c2 r03_c2 1000 1080 1920 32 16 > /dev/null
BENCH_ARGS="--window 64 --overlap 48" bash tools/profile.sh r03_c3 > gpurun_out/profile_c3.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r03_c3 r03_c3 1000 1080 1920 64 48 > /dev/null
BENCH_ARGS="--height 2160 --width 3840" bash tools/profile.sh r03_c4 > gpurun_out/profile_c4.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r03_c4 r03_c4 1000 2160 3840 32 16 > /dev/null
mkdir -p gpurun_out/r3h; cp profiles/r03_*_summary.json gpurun_out/r3h/
for c in c2 c3 c4; do grep -h "piv_" gpurun_out/prof_r03_$c/trace_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_r03_$c/trace.log; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('profiles/r03_*_summary.json')):
    d=json.load(open(f))
    for k,v in d['kernels'].items():
        print(f.split('/')[-1], k[:60], {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('hbm_traffic_bytes','hbm_fetch_bytes','hbm_write_bytes','valu_inst_per_simd_per_4cyc','valu_inst_per_wave')}, v.get('trace'))
PY
