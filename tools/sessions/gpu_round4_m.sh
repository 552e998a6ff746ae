#!/bin/bash
# round 4, session m: the per-pair "fit" records of the rescue pass through the power-of-two-width kernel (Pow2Pair): correctness
# (rescue tests, fuzz, full-size parity in bench), A/B against the generic kernel (LSPIV_RESCUE_GENERIC=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q -x --timeout 400 2>&1 | tail -3
for round in 1 2 3; do
  python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 6 --tag "c2 pow2"
  LSPIV_RESCUE_GENERIC=1 python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 6 --tag "c2 generic"
done
python tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 4 --tag "c3 pow2"
LSPIV_RESCUE_GENERIC=1 python tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 4 --tag "c3 generic"
python tools/ab_time.py --height 2160 --width 3840 --window 32 --overlap 16 --pairs 500 --reps 4 --tag "c4 pow2"
LSPIV_RESCUE_GENERIC=1 python tools/ab_time.py --height 2160 --width 3840 --window 32 --overlap 16 --pairs 500 --reps 4 --tag "c4 generic"
python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 5 --dtype f32 --tag "c2 f32 pow2"
cd /tmp && export TMPDIR=/tmp
for g in 0 1; do
  [ $g = 1 ] && export LSPIV_RESCUE_GENERIC=1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/t_$g -o x -- python $R/tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 10 > /dev/null 2>&1
  grep -h "rescue" $(find /tmp/t_$g -name "*kernel_stats.csv") | cut -c1-160
done
timeout 500 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['cpu_baseline']; print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline'].get('launch_ms_with_rescue_kernels'), {k:v for k,v in c.items() if k.startswith('parity') and not isinstance(v, dict)})"
