#!/bin/bash
# the committed profile sets (rounds 4 - 6) (per-timestep C2 / C3 / C4 / float32, the two ensemble kernels), the rescue cost of the
# ensemble finish, and the bench line -- usage: profile_all.sh <tag>   (tag r06 -> profiles/r06_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
cd $R
mkdir -p gpurun_out/${TAG}_out
python tools/ens_rescue_cost.py 1000 2>&1 | tail -4 | tee gpurun_out/${TAG}_out/ens_rescue_cost.log
bash tools/profile.sh ${TAG}_c2 > gpurun_out/${TAG}_out/profile_c2.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_${TAG}_c2 ${TAG}_c2 1000 1080 1920 32 16 > /dev/null
BENCH_ARGS="--window 64 --overlap 48" bash tools/profile.sh ${TAG}_c3 > gpurun_out/${TAG}_out/profile_c3.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_${TAG}_c3 ${TAG}_c3 1000 1080 1920 64 48 > /dev/null
BENCH_ARGS="--height 2160 --width 3840" bash tools/profile.sh ${TAG}_c4 > gpurun_out/${TAG}_out/profile_c4.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_${TAG}_c4 ${TAG}_c4 1000 2160 3840 32 16 > /dev/null
PROFILE_CMD="python $R/tools/f32_launch.py 1000 20" bash tools/profile.sh ${TAG}_f32 > gpurun_out/${TAG}_out/profile_f32.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_${TAG}_f32 ${TAG}_f32 1000 1080 1920 32 16 > /dev/null
bash tools/profile_ens.sh ${TAG} 32 16 64 48 > gpurun_out/${TAG}_out/profile_ens.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_${TAG}_ens32 ${TAG}_ens32 1000 1080 1920 32 16 > /dev/null
python tools/summarize_profile.py gpurun_out/prof_${TAG}_ens64 ${TAG}_ens64 1000 1080 1920 64 48 > /dev/null
cp profiles/${TAG}_*_summary.json gpurun_out/${TAG}_out/
python - "$TAG" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(f'profiles/{sys.argv[1]}_*_summary.json')):
    d = json.load(open(f))
    for k, v in d['kernels'].items():
        if 'walk' not in k: continue
        print(f.split('/')[-1], k[12:70], {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ('hbm_traffic_bytes', 'hbm_fetch_bytes', 'hbm_write_bytes', 'valu_inst_per_simd_per_4cyc')}, v.get('trace'))
PY
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${TAG}_out/bench.err > gpurun_out/${TAG}_out/bench.json
python - "$TAG" <<'PY'
import json, sys
d = json.load(open(f'gpurun_out/{sys.argv[1]}_out/bench.json')); c = d.get('cpu_baseline', {})
print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline'].get('traffic'), d['roofline']['frac'], d['config']['binary']['binary_hash_matches'])
print(c.get('value'), c.get('cores'), {k: v for k, v in c.items() if k.startswith('parity') and not isinstance(v, dict)})
print('ensemble parity', c.get('ensemble_parity'))
for o in d['config'].get('other_configs', []): print(o['workload'][:60], o['pairs_per_s'], o['launch_ms'], o.get('kernel_ms'), o['roofline']['frac'], o['roofline'].get('traffic'), o.get('final_fit_rescue'), o.get('finish_ms'))
print(d['config'].get('host_fed_pairs_per_s')); print(d['config'].get('camera_to_velocity_pairs_per_s'))
PY
