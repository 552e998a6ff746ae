#!/bin/bash
# Round 6 (dead end, the LSPIV_NORM_CHUNK switch was not kept): both normalize passes over n frames at a time -- see docs/history.md
cd $GRAFT_REPO_ROOT
for c in 0 16 32 48 64 100; do LSPIV_NORM_CHUNK=$c python tools/rows_launch.py normalize 30 201 2>&1 | grep "normalize:" | cut -c1-90 | sed "s/^/chunk $c /"; done
