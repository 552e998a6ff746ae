#!/bin/bash
# round 6, final tree: the long randomised differential runs (kernels, callers, ensemble mode, rows; narrow and wide grids)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/fuzz6
for s in ${SEEDS_A:-6001 6002 6003 6004}; do
  timeout 900 python tools/fuzz_parity.py $s 60 > gpurun_out/fuzz6/parity_$s.log 2>&1; echo "parity $s: $(tail -1 gpurun_out/fuzz6/parity_$s.log)"; grep FAIL gpurun_out/fuzz6/parity_$s.log | head -3
done
for s in ${SEEDS_B:-6011 6012}; do
  FUZZ_WIDE=1 timeout 900 python tools/fuzz_parity.py $s 20 > gpurun_out/fuzz6/parity_wide_$s.log 2>&1; echo "parity wide $s: $(tail -1 gpurun_out/fuzz6/parity_wide_$s.log)"; grep FAIL gpurun_out/fuzz6/parity_wide_$s.log | head -3
done
for s in ${SEEDS_C:-6021 6022 6023 6024}; do
  timeout 900 python tools/fuzz_modes.py $s 80 > gpurun_out/fuzz6/modes_$s.log 2>&1; echo "modes $s: $(tail -1 gpurun_out/fuzz6/modes_$s.log)"; grep -i FAIL gpurun_out/fuzz6/modes_$s.log | head -3
done
for s in ${SEEDS_D:-6031 6032 6033 6034}; do
  FUZZ_MODE=ensemble timeout 900 python tools/fuzz_modes.py $s 80 > gpurun_out/fuzz6/ens_$s.log 2>&1; echo "ensemble $s: $(tail -1 gpurun_out/fuzz6/ens_$s.log)"; grep -i FAIL gpurun_out/fuzz6/ens_$s.log | head -3
done
for s in ${SEEDS_E:-6041 6042 6043 6044}; do
  timeout 900 python tools/fuzz_rows.py $s 150 > gpurun_out/fuzz6/rows_$s.log 2>&1; echo "rows $s: $(tail -1 gpurun_out/fuzz6/rows_$s.log)"; grep -v "^ok" gpurun_out/fuzz6/rows_$s.log | head -3
done
