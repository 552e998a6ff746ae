#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r2g
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r2g/pytest.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err
( timeout 300 python tools/dtype_bench.py ) > gpurun_out/r2g/dtype.log 2>&1
( timeout 300 python tools/ensemble_bench.py 501 ) > gpurun_out/r2g/ensemble.log 2>&1
tail -3 gpurun_out/r2g/pytest.log; cat gpurun_out/r2g/dtype.log gpurun_out/r2g/ensemble.log | tail -12; python -c "
import json; d=json.load(open('gpurun_out/r2g/bench.json')); c=d['config']; print(d['value'], d['roofline']['kernel_ms_per_launch'], [o['pairs_per_s'] for o in c['other_configs']], c['camera_to_velocity_pairs_per_s']['device_resident_stages'], d['cpu_baseline']['parity_max_rel_err_vs_oracle'], d['cpu_baseline']['parity_nan_mismatch'])"
