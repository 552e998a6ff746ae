#!/bin/bash
# Round 6: blur_strip4_kernel (four columns per lane, integer row pass) over rows per strip / non-temporal stores, then every unrolled
# radius against the one-column kernel.  (First runs of this session: 1.0 - 1.3 ms whatever the strip height -- the first and the last
# strip of a row gathered their 12 bytes per lane by byte loads; with the loads clamped into the row and v_perm_b32 at the edge 0.42.)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out/blur4; mkdir -p $OUT
timeout 600 python -m pytest tests/test_filters.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
for rep in 1 2; do
for lib in tree blur4_ts24 blur4_ts48 blur4_ts64 blur4_nt onecol; do
  for row in smooth edge_detect; do
    case $lib in
      tree)   python tools/rows_launch.py $row 30 201;;
      onecol) LSPIV_BLUR_ONE_COLUMN=1 python tools/rows_launch.py $row 30 201;;
      *)      LSPIV_LIBRARY=build/ab/lib_$lib.so python tools/rows_launch.py $row 30 201;;
    esac | sed "s/^/$lib /" | cut -c1-110 | tee -a $OUT/ab.log
  done
done
done
cat > /tmp/radii.py <<'PY'
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyorc_amd import _lib
lib = _lib.load(); _lib.require_device()
H, W, T = 1080, 1920, 201
n = H * W
d_f, d_o = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * n)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), T * n * 4))
_lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, 3, 0.02))
def timed(fn, reps=20):
    for _ in range(3): fn()
    _lib.check(lib.lspiv_synchronize()); t0 = time.perf_counter()
    for _ in range(reps): fn()
    _lib.check(lib.lspiv_synchronize()); return (time.perf_counter() - t0) / reps
out = []
for k in (3, 5, 7): out.append("k=%d %.3f" % (k, 1e3 * timed(lambda: _lib.check(lib.lspiv_gaussian_blur_dev(d_f, 0, T, H, W, k, d_o, None)))))
for a, b in ((3, 5), (3, 7), (5, 7)): out.append("edge %d/%d %.3f" % (a, b, 1e3 * timed(lambda: _lib.check(lib.lspiv_edge_detect_dev(d_f, 0, T, H, W, a, b, d_o, None)))))
print(os.environ.get("TAG"), " | ".join(out))
PY
TAG=four-column python /tmp/radii.py | tee -a $OUT/ab.log
TAG=one-column LSPIV_BLUR_ONE_COLUMN=1 python /tmp/radii.py | tee -a $OUT/ab.log
