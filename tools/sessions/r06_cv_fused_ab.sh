#!/bin/bash
# Round 6: remap_fused_kernel, variants of build/ab against the tree's library (timing only: the knock-out variants compute nonsense)
cd $GRAFT_REPO_ROOT
for lib in tree $VARIANTS; do
  if [ $lib = tree ]; then python tools/rows_launch.py project_cv 30 201; else LSPIV_LIBRARY=build/ab/lib_$lib.so python tools/rows_launch.py project_cv 30 201; fi 2>&1 | grep "project_cv:" | cut -c1-100 | sed "s/^/$lib /"
done
