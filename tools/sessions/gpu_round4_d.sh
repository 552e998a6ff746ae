#!/bin/bash
# round 4, session d: the whole GPU suite on the staged ensemble finish + the sharding tests + the bench through pyorc_amd.shard
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4d
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 2>&1 | tail -40 > gpurun_out/r4d/suite.log
tail -25 gpurun_out/r4d/suite.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r4d/bench.err > gpurun_out/r4d/bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d/bench.json')); c=d.get('cpu_baseline',{})
print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline'].get('traffic'), d['roofline']['frac'], d['config'].get('binary'))
print(c.get('value'), {k:v for k,v in c.items() if k.startswith('parity') and not isinstance(v, dict)})
for o in d['config'].get('other_configs', []): print(o['workload'][:40], o['pairs_per_s'], o['launch_ms'], o['kernel_ms'])
print(d['config'].get('host_fed_pairs_per_s')); print(d['config'].get('camera_to_velocity_pairs_per_s'))
PY
LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --pairs 500 2> gpurun_out/r4d/bench2.err > gpurun_out/r4d/bench2.json; python -c "
import json; d=json.load(open('gpurun_out/r4d/bench2.json')); print(d['value'], d['ms_per_step'], json.dumps(d['config']['comm'])[:1500])"
