#!/bin/bash
# round 4, session u: the new tests (job-order invariance, bench with 4 / 8 ranks on one GPU), then randomised differential runs of
# every mode on the final binary
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4u
timeout 900 python -m pytest tests/test_gpu_strip_order.py tests/test_gpu_bench.py -m gpu -q --timeout 600 2>&1 | tail -6
for s in 601 602 603 604; do timeout 300 python tools/fuzz_parity.py $s 150 > gpurun_out/r4u/parity_$s.log 2>&1; echo "parity $s: $(tail -1 gpurun_out/r4u/parity_$s.log | cut -c1-200)"; grep FAIL gpurun_out/r4u/parity_$s.log | head -3; done
for m in timestep planes; do for s in 611 612 613 614; do FUZZ_MODE=$m timeout 300 python tools/fuzz_modes.py $s 60 > gpurun_out/r4u/${m}_$s.log 2>&1; echo "$m $s: $(grep -E 'cases,' gpurun_out/r4u/${m}_$s.log | tail -1 | cut -c1-200)"; grep FAIL gpurun_out/r4u/${m}_$s.log | head -3; done; done
timeout 300 python tools/fuzz_rows.py 621 100 2>&1 | tail -2
