# project_tile_kernel against project_mix_kernel over ortho grids of other oversampling ratios (1080p camera)
cd /root/repo
timeout 600 python -m pytest tests/test_project.py -m gpu -x -q 2>&1 | tail -5
for g in 810x1440 540x960 648x1152 1080x1920 1296x2304; do
  for r in project project_nn; do
    echo "== $g $r"
    ROWS_ORTHO=$g LSPIV_PROJECT_DEBUG=1 timeout 120 python tools/rows_launch.py $r 30 201 2>&1 | grep -E "tiles of|no mixed|^project" | cut -c1-110
    ROWS_ORTHO=$g LSPIV_PROJECT_NO_TILE=1 timeout 120 python tools/rows_launch.py $r 30 201 | cut -c1-110
  done
done
