#!/bin/bash
# round 3, closing session after the phase priorities (profiles first, on a cool GPU: the 32 x 32 kernels run the chip at its power limit and a box that has just run two minutes of other work clocks 10 % lower): GPU suite, the committed profile sets of the three single-GPU BASELINE configs, the full bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r3q
bash tools/profile.sh r03_c2 > gpurun_out/profile_c2.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r03_c2 r03_c2 1000 1080 1920 32 16 > /dev/null
BENCH_ARGS="--window 64 --overlap 48" bash tools/profile.sh r03_c3 > gpurun_out/profile_c3.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r03_c3 r03_c3 1000 1080 1920 64 48 > /dev/null
BENCH_ARGS="--height 2160 --width 3840" bash tools/profile.sh r03_c4 > gpurun_out/profile_c4.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r03_c4 r03_c4 1000 2160 3840 32 16 > /dev/null
PROFILE_CMD="python $R/tools/f32_launch.py 1000 20" bash tools/profile.sh r03_f32 > gpurun_out/profile_f32.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r03_f32 r03_f32 1000 1080 1920 32 16 > /dev/null
grep -o '"pairs_per_s": [0-9.]*' gpurun_out/prof_r03_f32/trace.log | tail -1
cp profiles/r03_*_summary.json gpurun_out/r3q/
for c in c2 c3 c4; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_r03_$c/trace.log; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('profiles/r03_*_summary.json')):
    d=json.load(open(f))
    for k,v in d['kernels'].items():
        print(f.split('/')[-1], k[:60], {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('hbm_traffic_bytes','hbm_fetch_bytes','hbm_write_bytes','valu_inst_per_simd_per_4cyc','valu_inst_per_wave')}, v.get('trace'))
PY
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r3q/bench.err | tee gpurun_out/r3q/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline'].get('launch_ms_with_rescue_kernels'), d['roofline'].get('traffic'), d['roofline']['frac'], d['config'].get('rescue'))
print(c['value'], {k:v for k,v in c.items() if k.startswith('parity') and not isinstance(v, dict)})
for o in d['config'].get('other_configs', []): print(o['workload'][:40], o['pairs_per_s'], o['launch_ms'], o['kernel_ms'], o['rescued_windows_per_launch'], o['roofline']['frac'], o['roofline'].get('traffic'))
print(d['config'].get('host_fed_pairs_per_s')); print(d['config'].get('camera_to_velocity_pairs_per_s'))
"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -3
