#!/bin/bash
# Round 6: blur_strip4f_kernel (float32 frames, four columns per lane, DPP neighbours) over rows per strip, against the one-column kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out/blur4f; mkdir -p $OUT
timeout 600 python -m pytest tests/test_filters.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
for rep in 1 2; do
for lib in tree b4f_ts8 b4f_ts24 b4f_ts32 onecol; do
  for row in smooth_f32 edge_detect_f32; do
    case $lib in
      tree)   python tools/rows_launch.py $row 30 201;;
      onecol) LSPIV_BLUR_ONE_COLUMN=1 python tools/rows_launch.py $row 30 201;;
      *)      LSPIV_LIBRARY=build/ab/lib_$lib.so python tools/rows_launch.py $row 30 201;;
    esac 2>&1 | grep "_f32:" | sed "s/^/$lib /" | cut -c1-115 | tee -a $OUT/ab.log
  done
done
done
