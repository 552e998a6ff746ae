#!/bin/bash
# round 4, session n: partial-sum kernels of the ensemble rescue, power-of-two-width variant against the generic one, with the flag
# allowance raised so that hundreds of windows are flagged (LSPIV_RESCUE_KAPPA)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 1000 4000; do
  echo "kappa $k pow2";    LSPIV_RESCUE_KAPPA=$k python tools/ens_rescue_cost.py 1000 2>&1 | grep "rescue 1"
  echo "kappa $k generic"; LSPIV_RESCUE_KAPPA=$k LSPIV_RESCUE_GENERIC=1 python tools/ens_rescue_cost.py 1000 2>&1 | grep "rescue 1"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "rescue or ensemble" --timeout 300 2>&1 | tail -2
