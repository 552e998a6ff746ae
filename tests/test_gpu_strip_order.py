"""The walking kernels take their (segment, window) jobs in column strips for some shapes (csrc/piv_fft_impl.h, strip_order: 32 windows
for 64 x 64, 24 for 32 x 32 windows of float32 / float64 frames) -- an ORDER of the same jobs, so the results must not depend on it,
bit for bit: default against row-major (LSPIV_STRIP_W=0) and an odd width that leaves a remainder strip, grids wider and narrower
than a strip, per time step and ensemble."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tmp_path, tag, strip_w, args):
    env = dict(os.environ)
    env.pop("LSPIV_STRIP_W", None)
    if strip_w is not None:
        env["LSPIV_STRIP_W"] = str(strip_w)
    out = os.path.join(tmp_path, f"{tag}.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "strip_order_worker.py"), out] + [str(x) for x in args],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


@pytest.mark.parametrize("args", [
    (32, 16, "float32", 400, 1000, 30, 0),     # 61 columns: two strips of 24 + a remainder of 13
    (32, 16, "float64", 300, 340, 27, 0),      # 20 columns: narrower than a strip
    (64, 48, "uint8", 300, 1200, 26, 0),       # 72 columns: two strips of 32 + 8
    (64, 48, "uint8", 300, 1200, 30, 1),
    (32, 16, "float32", 300, 900, 30, 1),
    (64, 48, "uint8", 200, 1200, 6, 2),        # ... and the plane volume
    (32, 16, "float32", 200, 900, 7, 2),
    (64, 48, "uint8", 300, 1200, 26, 0, 0.3),  # the signal-threshold variants of the kernels
    (64, 48, "uint8", 300, 1200, 30, 1, 0.3),
    (32, 16, "float32", 300, 900, 30, 1, 0.3),
    (32, 16, "uint8", 300, 900, 30, 0),        # row-major by default: a strip order forced on it changes nothing either
])
def test_results_do_not_depend_on_the_job_order(gpu, tmp_path, args):
    ref = run(tmp_path, "default", None, args)
    for sw in (0, 7):
        got = run(tmp_path, f"w{sw}", sw, args)
        for k in ref.files:
            assert np.array_equal(ref[k].view(np.uint32), got[k].view(np.uint32)), (sw, k)
