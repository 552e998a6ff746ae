#!/bin/bash
# round 3: interleaved A/B of library builds (name=path ...) on C2 only, 4 rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2 3 4; do
  for kv in "$@"; do
    LSPIV_LIBRARY=${kv#*=} timeout 120 python tools/ab_time.py --tag ${kv%%=*} 2>&1 | tail -1
  done
done
