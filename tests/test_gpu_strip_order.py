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


def run(tmp_path, tag, strip_w, args, xcd_order=None):
    env = dict(os.environ)
    env.pop("LSPIV_STRIP_W", None)
    env.pop("LSPIV_XCD_ORDER", None)
    if strip_w is not None:
        env["LSPIV_STRIP_W"] = str(strip_w)
    if xcd_order is not None:
        env["LSPIV_XCD_ORDER"] = str(xcd_order)
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
    # ... nor on how the jobs are dealt to the XCDs (round 5: an eighth of the windows of every segment per XCD by default;
    # LSPIV_XCD_ORDER=0: one contiguous range of jobs per XCD, as in rounds 2 - 4), alone and together with another strip width
    for sw, xo in ((None, 0), (7, 0)):
        got = run(tmp_path, f"x{xo}w{sw}", sw, args, xcd_order=xo)
        for k in ref.files:
            assert np.array_equal(ref[k].view(np.uint32), got[k].view(np.uint32)), (sw, xo, k)


@pytest.mark.parametrize("ws,ov,ens", [(32, 16, 0), (64, 48, 0), (32, 16, 1), (64, 48, 1)])
def test_long_anchors_on_the_full_1080p_grid(gpu, ws, ov, ens):
    """Round 5: on grids with at least as many windows as the chip has lane groups the walking kernels' segments are 75 pairs long
    (lspiv_chunk_alignment_grid).  At 1080p (7 854 / 7 488 windows), 290 pairs resident in HBM: chunks cut on multiples of 75 -- with a
    ragged end -- reproduce one launch bit for bit, per time step and in ensemble mode; a chunk cut on a multiple of 25 that is no
    multiple of 75 is still correct but only equal to rounding (which is why the planner cuts on 75 there)."""
    import ctypes as C

    from pyorc_amd import _lib, piv, window

    lib = _lib.load()
    H, W, P = 1080, 1920, 290
    assert window.chunk_alignment((ws, ws), (H, W), (ov, ov)) == 75 and window.chunk_alignment((ws, ws), (128, 160), (ws // 2, ws // 2)) == 25
    nr, nc = window.get_array_shape((H, W), (ws, ws), (ov, ov))
    n_win = nr * nc
    d_f, d_o = C.c_void_p(), C.c_void_p()
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_f), (P + 1) * H * W)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * n_win))
    _lib.check(lib.lspiv_synth_particles_dev(d_f, P + 1, H, W, 77, 0.02))
    at = lambda f: C.c_void_p(d_f.value + f * H * W)

    def per_timestep(bounds):
        out = np.empty((4, P, n_win), np.float32)
        for a, b in zip(bounds, bounds[1:]):
            _lib.check(lib.lspiv_piv_pairs_dev_at(at(a), 0, b - a + 1, H, W, ws, ws, ov, ov, -1.0, a, d_o, None, None))
            blk = np.empty((4, b - a, n_win), np.float32)
            _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(blk), d_o, blk.nbytes))
            out[:, a:b] = blk
        return out

    def ensemble(bounds):
        e = piv.Ensemble((H, W), (ws, ws), (ov, ov))
        e.set_retain(e.RETAIN_BORROW)
        for a, b in zip(bounds, bounds[1:]):
            e.accumulate_dev(at(a).value, np.uint8, b - a + 1, 0.2, 3.0, d_o.value)
        s, k = e.export_state()
        u, v, cnt = e.finish(0.2, 1)
        e.close()
        return s, k, u, v

    run_ = ensemble if ens else per_timestep
    whole = run_([0, P])
    for bounds in ([0, 75, 225, P], [0, 150, P], [0, 225, P]):
        got = run_(bounds)
        for x, y in zip(whole, got) if ens else ((whole, got),):
            assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)), bounds
    off = run_([0, 100, P])          # cut on the family's base length only: the same results up to float32 rounding
    for x, y in zip(whole, off) if ens else ((whole, off),):
        x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
        assert np.array_equal(np.isnan(x), np.isnan(y))
        assert np.nanmax(np.abs(x - y) / np.maximum(np.abs(x), 0.05 if not ens else 1.0), initial=0.0) < 2e-3
    lib.lspiv_dev_free(d_f); lib.lspiv_dev_free(d_o)
