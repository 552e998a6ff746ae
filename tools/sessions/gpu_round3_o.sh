#!/bin/bash
# round 3, last session: the GPU suite, smoke(), bench.py under torch.distributed.run (two ranks sharing the one GPU over the
# shared-memory transport: the launcher path the driver uses for N > 1), and the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 3 --warmup 1 --pairs 200 > gpurun_out/r3o/bench_torchrun2.json 2> gpurun_out/r3o/bench_torchrun2.err
echo "torchrun rc $?"; python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r3o/bench_torchrun2.json') if l.strip().startswith('{')][-1])
    print('torchrun 2 ranks:', d['n_gpus'], d['value'], d['scaling'], d['config']['comm'])
except Exception as e:
    print('torchrun parse failed', e); print(open('gpurun_out/r3o/bench_torchrun2.err').read()[-2000:])
PY
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r3o/bench.err | tee gpurun_out/r3o/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline'].get('launch_ms_with_rescue_kernels'), d['roofline'].get('traffic'), d['roofline']['frac'], d['config'].get('rescue'))
print(c['value'], {k:v for k,v in c.items() if k.startswith('parity') and not isinstance(v, dict)})
for o in d['config'].get('other_configs', []): print(o['workload'][:40], o['pairs_per_s'], o['launch_ms'], o['kernel_ms'], o['rescued_windows_per_launch'], o['roofline']['frac'], o['roofline'].get('traffic'))
print(d['config'].get('host_fed_pairs_per_s')); print(d['config'].get('camera_to_velocity_pairs_per_s'))
"
