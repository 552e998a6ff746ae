"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Gate: max |delta| / max(|ref|, 0.05) <= 1e-4 for u, v (pixels), corr_max and s2n, NaN masks
identical -- on ALL windows since round 3: the kernels flag the windows whose float32 sub-pixel fit is ill-conditioned
(a neighbour of the peak that is zero in exact arithmetic, a flat ridge, two samples tying for the maximum) and the
float64 rescue pass (csrc/piv_rescue.hip) re-evaluates them from the frames.  Only exact float64 ties of the plane
maximum are set aside (oracle.c_oracle.exact_tie): there the oracle's own pick is a matter of its FFT's rounding.
"""
import ctypes as C
import json
import os

import warnings

import numpy as np
import pytest

from oracle import c_oracle
from oracle import piv_oracle as po
from pyorc_amd.synth import flow_field, particle_stack

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "piv_golden.npz"))
TOL = 1e-4


def rel_err(got, ref, floor=0.05):
    with np.errstate(all="ignore"):
        e = np.abs(np.asarray(got, dtype=np.float64) - ref) / np.maximum(np.abs(ref), floor)
    return float(np.nanmax(e)) if np.isfinite(e).any() else 0.0


# Which windows are set aside is decided by the ORACLE's own float64 planes (c_oracle.exact_tie on the oracle's outputs), never by
# anything the kernel returned -- a kernel cannot hide windows here.  The share that is gated is nevertheless pinned (VERDICT r04 item
# 8): tests/golden/tie_shares.json holds, per test and call, the share measured when the fixture was recorded (the degenerate inputs
# of some tests -- empty frames, constant corners -- put up to 31 % of their windows into exact ties); a comparison must reproduce its
# recorded share to 0.02, and one without a record must gate at least 90 % of its windows.
_TIE_SHARES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tie_shares.json")))
_tie_calls = {}


def _check_tie_share(share, n):
    _log_tie_share(share, n)
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0].split("/")[-1]
    k = _tie_calls[test] = _tie_calls.get(test, -1) + 1
    rec = _TIE_SHARES.get(test)
    if rec is not None and k < len(rec):
        assert abs(share - rec[k]) <= 0.02, f"{test} call {k}: {share:.4f} of the windows gated, {rec[k]:.4f} when the fixture was recorded"
    else:
        assert share >= 0.9, f"{test} call {k}: only {share:.4f} of the windows gated and no recorded expectation (tools/record_tie_shares.py)"


def _log_tie_share(share, n):
    """LSPIV_TIE_LOG=<file>: one line per gated comparison -- the test, the share of windows that were gated (1 - exact ties)."""
    path = os.environ.get("LSPIV_TIE_LOG")
    if path:
        with open(path, "a") as fh:
            fh.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?')}\t{share:.6f}\t{n}\n")


def check_against_oracle(frames, ws, ov, thr=None, plane_tol=2e-6, uv_tol=TOL):
    """GPU vs oracle on ALL windows: NaN masks, corr / s2n, planes, and u, v to 1e-4 of max(|ref|, 0.05 px) -- the float64
    rescue pass (csrc/piv_rescue.hip) covers the windows whose float32 fit is ill-conditioned.  Only exact float64 ties of
    the plane maximum are set aside (c_oracle.exact_tie)."""
    import pyorc_amd

    u, v, cm, sn, planes = pyorc_amd.piv_pairs(frames, ws, ov, thr, return_planes=True)
    uo, vo, cmo, sno, po_planes, cond = c_oracle.piv_pairs(frames, ws, ov, thr, return_planes=True, return_cond=True)
    ok = ~c_oracle.exact_tie(cond, cmo)
    assert u.dtype == v.dtype == cm.dtype == sn.dtype == np.float32
    assert u.shape == uo.shape
    for name, g, r in (("u", u, uo), ("v", v, vo)):
        assert np.array_equal(np.isnan(g)[ok], np.isnan(r)[ok]), f"{name}: NaN mask differs"
    for name, g, r in (("corr", cm, cmo), ("s2n", sn, sno)):
        assert np.array_equal(np.isnan(g), np.isnan(r)), f"{name}: NaN mask differs"
    assert rel_err(cm, cmo.astype(np.float64)) <= TOL
    assert rel_err(sn, sno.astype(np.float64)) <= TOL
    assert np.array_equal(np.isnan(planes), np.isnan(po_planes))
    assert np.nanmax(np.abs(planes - po_planes), initial=0.0) < plane_tol
    if ok.any():
        assert rel_err(u[ok], uo[ok].astype(np.float64)) <= uv_tol
        assert rel_err(v[ok], vo[ok].astype(np.float64)) <= uv_tol
    _check_tie_share(float(ok.mean()), ok.size)
    return u, v, cm, sn


# ------------------------------------------------------------------ golden fixtures -----------------
def test_g1_known_shifts(gpu):
    import pyorc_amd

    for fr, exp in zip(GOLD["g1_frames"], GOLD["g1_expected"]):
        u, v, cm, sn = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16))
        got = np.array([u[0, 0, 0], v[0, 0, 0], cm[0, 0, 0], sn[0, 0, 0]])
        assert rel_err(got, exp.astype(np.float64)) <= TOL


def test_g2_degenerate_windows(gpu):
    import pyorc_amd

    for name, fr, exp in zip(GOLD["g2_names"], GOLD["g2_frames"], GOLD["g2_expected"]):
        u, v, cm, sn = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16))
        got = np.array([u[0, 0, 0], v[0, 0, 0], cm[0, 0, 0], sn[0, 0, 0]])
        assert np.array_equal(np.isnan(got), np.isnan(exp)), name
        if name == "single_px":
            # a one-sample spike on an exactly-zero plane: the neighbours of the peak are 0 + 1e-7 in exact
            # arithmetic: the float32 fit amplifies ~1e-8 of rounding noise there, the kernel flags it for the rescue pass
            assert rel_err(got, exp.astype(np.float64)) <= TOL, name   # since round 3: the float64 rescue pass
            continue
        assert rel_err(got, exp.astype(np.float64)) <= TOL, name
        if name in ("const_a", "zero_b", "both_zero"):
            assert got[2] == 0.0  # exactly zero plane, not rounding noise


@pytest.mark.parametrize("tag,ws,ov", [("u8", (32, 32), (16, 16)), ("u8", (64, 64), (48, 48)),
                                       ("f32", (32, 32), (16, 16)), ("f32", (64, 64), (48, 48))])
def test_g3_mini_stack(gpu, tag, ws, ov):
    import pyorc_amd

    u, v, cm, sn = pyorc_amd.piv_pairs(GOLD[f"g3_frames_{tag}"], ws, ov)
    for k, got in (("u", u), ("v", v), ("corr", cm), ("s2n", sn)):
        exp = GOLD[f"g3_{tag}_{ws[0]}_{k}"]
        assert np.array_equal(np.isnan(got), np.isnan(exp)), k
        assert rel_err(got, exp.astype(np.float64)) <= TOL, k


def test_g3_signal_threshold(gpu):
    import pyorc_amd

    u, v, cm, sn = pyorc_amd.piv_pairs(GOLD["g3_frames_u8"], (32, 32), (16, 16), signal_threshold=0.3)
    for k, got in (("u", u), ("v", v), ("corr", cm), ("s2n", sn)):
        exp = GOLD[f"g3_u8_32_thr03_{k}"]
        assert np.array_equal(np.isnan(got), np.isnan(exp)), k
        assert rel_err(got, exp.astype(np.float64)) <= TOL, k


# ------------------------------------------------------------------ oracle on seeded inputs ---------
@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
@pytest.mark.parametrize("ws,ov", [((32, 32), (16, 16)), ((32, 32), (24, 8)), ((32, 32), (0, 0))])
def test_fft32_kernel_vs_oracle(gpu, dtype, ws, ov):
    fr = particle_stack(4, 160, 224, seed=3)
    fr = fr if dtype == np.uint8 else (fr.astype(dtype) - 31.25) * 0.5  # signed, non-integer
    check_against_oracle(fr, ws, ov)


@pytest.mark.parametrize("shape", [(157, 211), (33, 47), (32, 32), (65, 1025)])
def test_ragged_and_unaligned_frames(gpu, shape):
    fr = particle_stack(3, shape[0], shape[1], seed=9, density=0.04)
    check_against_oracle(fr, (32, 32), (16, 16))
    check_against_oracle(fr.astype(np.float32), (32, 32), (16, 16))


@pytest.mark.parametrize("ws,ov", [((10, 10), (5, 5)), ((24, 24), (12, 12)), ((24, 16), (12, 8)), ((26, 26), (13, 13)),
                                   ((64, 64), (48, 48)), ((16, 16), (8, 8)), ((4, 6), (2, 3))])
def test_other_window_sizes_vs_oracle(gpu, ws, ov):
    fr = particle_stack(3, 128, 144, seed=17, density=0.06)
    check_against_oracle(fr, ws, ov)


@pytest.mark.parametrize("ws,ov,shape", [((32, 32), (31, 31), (40, 45)), ((32, 32), (31, 0), (34, 100)),
                                         ((64, 64), (63, 60), (70, 80)), ((12, 12), (11, 11), (20, 24))])
def test_extreme_overlaps(gpu, ws, ov, shape):
    """Stride-1 grids (overlap = window - 1): every pixel shift is its own window; exercises odd grids and the
    window-index arithmetic of all three kernels."""
    fr = particle_stack(3, shape[0], shape[1], seed=12, density=0.05)
    check_against_oracle(fr, ws, ov)


def test_input_layouts_and_dtypes(gpu):
    """Whatever numpy hands over: non-contiguous views, Fortran order, small integer types, bool, T = 2."""
    import pyorc_amd

    base = particle_stack(4, 96, 128, seed=14)
    ref = pyorc_amd.piv_pairs(base, (32, 32), (16, 16))
    wide = np.zeros((4, 96, 256), np.uint8); wide[:, :, ::2] = base
    for view in (wide[:, :, ::2], np.asfortranarray(base), base.astype(np.int16), base.astype(np.uint16),
                 base.astype(np.float64), base.astype(np.float32)):
        got = pyorc_amd.piv_pairs(view, (32, 32), (16, 16))
        for a, b in zip(ref, got):
            assert np.array_equal(np.isnan(a), np.isnan(b)) and rel_err(b, a.astype(np.float64)) <= TOL
    two = pyorc_amd.piv_pairs(base[:2], (32, 32), (16, 16))
    assert two[0].shape == (1, 5, 7) and np.array_equal(two[0][0], ref[0][0], equal_nan=True)
    b = pyorc_amd.piv_pairs(base > 40, (32, 32), (16, 16))
    uo, vo, cmo, sno = c_oracle.piv_pairs((base > 40).astype(np.float64), (32, 32), (16, 16))
    assert rel_err(b[2], cmo.astype(np.float64)) <= TOL
    check_against_oracle(base.astype(np.float64) - 9.5, (64, 64), (48, 48), thr=0.2)


def test_signal_threshold_vs_oracle(gpu):
    fr = particle_stack(4, 160, 224, seed=5)
    fr[:, :, :64] = 0  # empty band: skipped windows
    for thr in (0.0, 0.2, 0.35, 1.0):
        u, *_ = check_against_oracle(fr, (32, 32), (16, 16), thr=thr)
    assert np.isnan(u).all()  # threshold 1.0 skips everything
    check_against_oracle((fr.astype(np.float32) - 3.0), (32, 32), (16, 16), thr=0.9)


def test_constant_and_empty_windows(gpu):
    fr = particle_stack(3, 128, 160, seed=23)
    fr[:, :40, :40] = 7
    fr[:, 90:, 100:] = 0
    u, v, cm, sn = check_against_oracle(fr, (32, 32), (16, 16))
    assert cm[0, 0, 0] == 0.0 and np.isnan(sn[0, 0, 0]) and np.isnan(u[0, 0, 0])
    f = fr.astype(np.float32) * 0.1  # constant 0.7 windows: the mean must come out exactly
    u, v, cm, sn = check_against_oracle(f, (32, 32), (16, 16))
    assert cm[0, 0, 0] == 0.0


def test_non_finite_input_gives_nan_not_garbage(gpu):
    import pyorc_amd

    fr = particle_stack(2, 64, 64, seed=1).astype(np.float32)
    fr[1, 10, 10] = np.nan
    u, v, cm, sn = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16))
    assert np.isnan(u[0, 0, 0]) and np.isnan(cm[0, 0, 0]) and np.isnan(sn[0, 0, 0])
    assert np.isfinite(cm[0, 1, 1])  # window (1,1) starts at (16,16) and contains the NaN too -> NaN
    fr[1, 10, 10] = np.inf
    u, v, cm, sn = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16))
    assert np.isnan(cm[0, 0, 0])


def test_flow_recovery_sanity(gpu):
    import pyorc_amd
    from pyorc_amd import window

    fr = particle_stack(3, 256, 320, seed=11)
    u, v, cm, sn = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16))
    x, y = window.get_rect_coordinates(fr.shape[1:], (32, 32), (16, 16))
    ut, vt = flow_field(256, 320, y[:, None].astype(float), x[None, :].astype(float))
    assert np.nanmedian(np.abs(u[0] - ut)) < 0.15 and np.nanmedian(np.abs(v[0] - vt)) < 0.15


# ------------------------------------------------------------------ ffpiv-shaped API ----------------
def test_cross_corr_and_u_v_displacement_drop_ins(gpu):
    import pyorc_amd

    fr = GOLD["g3_frames_u8"]
    x, y, corr = pyorc_amd.cross_corr(fr, window_size=(32, 32), overlap=(16, 16), search_area_size=(32, 32),
                                      normalize=False, engine="hip", signal_threshold=None, verbose=False)
    xo, yo, co = po.cross_corr(fr, (32, 32), (16, 16))
    assert np.array_equal(x, xo) and np.array_equal(y, yo)
    assert corr.shape == co.shape and corr.dtype == np.float32
    assert np.abs(corr - co).max() < 2e-6
    # the reference's reductions on top of the returned volume (pyorc/velocimetry/ffpiv.py:465-466)
    cmax = np.nanmax(corr, axis=(-1, -2))
    assert rel_err(cmax, np.nanmax(co, axis=(-1, -2))) <= TOL
    u, v = pyorc_amd.u_v_displacement(corr, len(y), len(x), engine="hip")
    uo, vo = po.u_v_displacement(co, len(yo), len(xo))
    assert np.array_equal(np.isnan(u), np.isnan(uo))
    assert rel_err(u, uo) <= TOL and rel_err(v, vo) <= TOL
    # planes from any source: NaN planes and border peaks give NaN, like np.argmax + peak_position
    vol = co.astype(np.float32).copy()
    vol[0, 0] = np.nan
    vol[0, 1] = 0.0
    vol[0, 1, 0, 5] = 1.0
    u2, v2 = pyorc_amd.u_v_displacement(vol, len(y), len(x))
    uo2, vo2 = po.u_v_displacement(vol.astype(np.float64), len(y), len(x))
    assert np.isnan(u2[0, 0, 0]) and np.isnan(u2[0, 0, 1]) and np.array_equal(np.isnan(u2), np.isnan(uo2))
    with pytest.raises(ValueError):
        pyorc_amd.cross_corr(fr, (32, 32), (16, 16), engine="numba")
    with pytest.raises(ValueError):
        pyorc_amd.u_v_displacement(corr, 3, 3)


def test_fused_results_equal_two_step_results(gpu):
    """piv_pairs (fused) == u_v_displacement(cross_corr(...)) + reductions, bit for bit on the GPU."""
    import pyorc_amd

    fr = particle_stack(3, 128, 160, seed=2)
    for ws, ov in (((32, 32), (16, 16)), ((24, 24), (12, 12))):
        u, v, cm, sn = pyorc_amd.piv_pairs(fr, ws, ov)
        x, y, corr = pyorc_amd.cross_corr(fr, ws, ov)
        u2, v2 = pyorc_amd.u_v_displacement(corr, len(y), len(x))
        assert np.array_equal(u, u2, equal_nan=True) and np.array_equal(v, v2, equal_nan=True)
        assert np.array_equal(cm.ravel(), corr.max(axis=(-1, -2)).ravel())


# ------------------------------------------------------------------ windows above 64 px --------------
@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
@pytest.mark.parametrize("ws,ov", [((96, 96), (48, 48)), ((128, 128), (64, 64)), ((128, 128), (96, 32)), ((66, 66), (33, 33)),
                                   ((100, 100), (50, 50)), ((97, 97), (40, 40)), ((128, 80), (64, 40)), ((72, 120), (0, 60)),
                                   ((72, 72), (36, 36)), ((80, 80), (40, 40)), ((84, 84), (42, 42)), ((90, 90), (45, 45)),
                                   ((112, 112), (56, 56)), ((120, 120), (60, 60)),
                                   ((41, 41), (20, 20)), ((63, 63), (31, 31)), ((64, 32), (32, 16)), ((49, 33), (10, 30)),
                                   # composite lengths in two passes: 7 x 14, 2 x 37, 6 x 19 and 9 x 14 (their padded tiles need
                                   # two rounds), 5 x 25, 10 x 11 with 8 x 11
                                   ((98, 98), (49, 49)), ((74, 74), (37, 37)), ((114, 114), (57, 57)), ((126, 126), (63, 63)),
                                   ((125, 125), (60, 60)), ((110, 88), (55, 44))])
def test_windows_above_64_vs_oracle(gpu, ws, ov, dtype):
    """ffpiv.cross_corr takes any window (pyorc/api/frames.py:159-168): sizes above 64 px -- 96 and 128 for 4K footage,
    but also odd, non-square and 2 x prime sizes -- run the LDS-resident DFT kernel; planes, NaN masks, corr / s2n and
    the sub-pixel peaks against the oracle, with a signal threshold, an empty frame and a constant corner in the stack.
    Square sizes N = R x M with a register FFT of length M (72 ... 128: 3 x 24, 4 x 20, 3 x 28, 3 x 30, 3 x 32, 5 x 20, 4 x 28,
    4 x 30, 4 x 32) take the four-step passes, everything else the DFT passes (a composite length as two shorter passes);
    LSPIV_NO_FOURSTEP=1 cross-checks.
    The DFT passes also serve the non-square and odd windows below 64 px from 1500 samples on (41 x 41, 63 x 63, 64 x 32)."""
    H, Wd = 2 * ws[0] + 7, 2 * ws[1] + ws[1] // 2 + 3
    fr = particle_stack(4, H, Wd, seed=ws[0] + ws[1], density=0.03)
    # planes out of the LDS-resident transforms meet the 2e-6 of the register FFT kernels since the windows are transformed
    # de-meaned (4e-6 before); the sub-pixel gate takes the windows whose peak neighbours reach 5 % of the maximum -- the
    # fuzz tool's rule for every size
    kw = dict(plane_tol=2e-6)
    if dtype == np.uint8:
        check_against_oracle(fr, ws, ov, **kw)
        check_against_oracle(fr, ws, ov, thr=0.12, **kw)
    else:
        f = (fr.astype(dtype) - 21.5) * 0.37
        f[2] = 1.5                                       # a constant frame: both pairs that touch it are dead
        f[:, : ws[0] // 2, : ws[1]] = -0.75              # a constant corner
        check_against_oracle(f, ws, ov, thr=0.25, **kw)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("ws,ov", [((160, 160), (80, 80)), ((256, 256), (128, 192)), ((130, 130), (65, 65)), ((192, 144), (96, 72)),
                                   ((64, 320), (32, 160)), ((251, 251), (100, 100))])
def test_windows_above_128_vs_oracle(gpu, ws, ov, dtype):
    """ffpiv.cross_corr has no upper bound on the window.  Two planes of more than 128 x 128 complex samples do not fit the
    LDS of a CU; those windows run the same transforms on slots of HBM scratch (kernel kind 10): composite sides in two
    passes and rounds (160 = 10 x 16, 256 = 16 x 16, 130 = 10 x 13, 192 x 144), a prime side (251) in single passes, a
    side below 65 next to one above 128 -- planes, NaN masks, corr / s2n and peaks against the oracle, plus ensemble mode."""
    import pyorc_amd.piv as P
    from pyorc_amd import _lib

    assert _lib.load().lspiv_kernel_kind(*ws) == 10
    H, Wd = 2 * ws[0] + 9, 2 * ws[1] + ws[1] // 3 + 5
    fr = particle_stack(4, H, Wd, seed=ws[0] + 3 * ws[1], density=0.03)
    # planes agree to 4e-7; the broad peaks of these windows (log-curvature down to 0.2) turn that into up to 6e-6 px, which is
    # 1.2e-4 of the 0.05-px floor of the relative measure at 256 x 256: the gate on u, v is 2e-4 here
    kw = dict(plane_tol=2e-6)
    if dtype == np.uint8:
        check_against_oracle(fr, ws, ov, **kw)
        check_against_oracle(fr, ws, ov, thr=0.12, **kw)
    else:
        f = (fr.astype(dtype) - 21.5) * 0.37
        f[2] = 1.5                                       # a constant frame: both pairs that touch it are dead
        f[:, : ws[0] // 2, : ws[1]] = -0.75              # a constant corner
        check_against_oracle(f, ws, ov, thr=0.25, **kw)
    if dtype == np.uint8 and ws[0] == ws[1]:             # ensemble mode through the same slots
        ens = P.Ensemble((H, Wd), ws, ov)
        ens.accumulate(fr, 0.05, 1.0)
        u, v, cnt = ens.finish(0.0, 3)
        ens.close()
        ref = po.get_ffpiv(fr, np.ones(3), ws, ov, 1.0, 1.0, ensemble_corr=True, corr_min=0.05, s2n_min=1.0, count_min=0.0)
        assert np.array_equal(np.isnan(u[0]), np.isnan(ref["v_x"][0])) and rel_err(u[0], ref["v_x"][0].astype(np.float64)) <= TOL
        assert rel_err(v[0], ref["v_y"][0].astype(np.float64)) <= TOL


@pytest.mark.parametrize("ws", [96, 128])
def test_windows_above_64_get_piv_and_ensemble(gpu, ws):
    """96 / 128 px windows through the accessor-shaped entry point: per-timestep (chunked: per-pair kernel, alignment 1)
    and ensemble mode against the oracle's get_ffpiv."""
    from pyorc_amd import frames as F
    from pyorc_amd import window

    assert window.chunk_alignment((ws, ws)) == 1
    fr = particle_stack(6, 2 * ws + 20, 3 * ws, seed=ws, density=0.03)
    t = np.arange(6) / 30.0
    got = F.get_piv(fr, ws, time=t, resolution=0.01)
    part = F.get_piv(fr, ws, time=t, resolution=0.01, chunksize=2)
    ref = po.get_ffpiv(fr, np.diff(t), (ws, ws), (ws // 2, ws // 2), 0.01, 0.01)
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(got[k], part[k], equal_nan=True), k
        assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), k
    assert rel_err(got["corr"], ref["corr"].astype(np.float64)) <= TOL and rel_err(got["s2n"], ref["s2n"].astype(np.float64)) <= TOL
    floor = 0.05 * 0.01 * 30
    assert rel_err(got["v_x"], ref["v_x"].astype(np.float64), floor=floor) <= TOL
    assert rel_err(got["v_y"], ref["v_y"].astype(np.float64), floor=floor) <= TOL
    ens = F.get_piv(fr, ws, time=t, resolution=0.01, ensemble_corr=True, corr_min=0.1, s2n_min=1.5)
    eref = po.get_ffpiv(fr, np.diff(t), (ws, ws), (ws // 2, ws // 2), 0.01, 0.01, ensemble_corr=True, corr_min=0.1, s2n_min=1.5)
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(np.isnan(ens[k]), np.isnan(eref[k])), k
    assert rel_err(ens["corr"], eref["corr"].astype(np.float64)) <= TOL
    assert rel_err(ens["v_x"], eref["v_x"].astype(np.float64), floor=floor) <= TOL


# ------------------------------------------------------------------ unpinned engine semantics (A5 / A7) ---
@pytest.mark.parametrize("ws", [(32, 32), (64, 64), (24, 24), (16, 16), (9, 9), (27, 27), (24, 16)])
@pytest.mark.parametrize("opt,val", [("border_peak", 1), ("border_peak", 2), ("signal_mode", 1), ("signal_positive", 1)])
def test_unpinned_semantics_switches(gpu, opt, val, ws):
    """The readings of ffpiv that /root/reference cannot decide (border peak -> NaN | centre | integer peak;
    signal_threshold per window pair | per window position over the chunk; non-zero | above zero) exist as options in
    the HIP library and as `semantics` in the numpy oracle; every alternative is checked here, on every kernel family
    (fused FFT 32 / 64, prime-factor, 16-point, embedded 32 / 64, direct), per-timestep, through the plane volume and in
    ensemble mode -- so that matching a real ffpiv run is a default flip, not new code."""
    import pyorc_amd
    import pyorc_amd.piv as P

    H, Wd = 3 * ws[0] + 5, 4 * ws[1] + 3
    fr = particle_stack(5, H, Wd, seed=50 + ws[0], density=0.05).astype(np.float32)
    fr -= 40.0                                          # signed samples: "non-zero" and "above zero" differ
    fr[fr < -35.0] = 0.0                                # exact zeros: the background carries no signal
    fr[:, : ws[0] + 2, : ws[1] + 2] = 0.0               # an empty corner: dropped by any threshold
    fr[2, :, Wd // 2:] = 0.0                            # one half-empty frame: pair mode drops 2 pairs, stack mode may not
    # one window whose content is rolled circularly by half a window between frames: its correlation peak sits at
    # lag -w/2, i.e. on the plane border (and several of its neighbours' peaks do as well)
    y0, x0 = ws[0] - ws[0] // 2, 2 * (ws[1] - ws[1] // 2)
    for t in range(1, 5):
        fr[t, y0:y0 + ws[0], x0:x0 + ws[1]] = np.roll(fr[t - 1, y0:y0 + ws[0], x0:x0 + ws[1]], ws[1] // 2, axis=1)
    thr = 0.3
    ov = (ws[0] // 2, ws[1] // 2)
    pyorc_amd.set_option(opt, val)
    try:
        with po.semantics(**{opt: val}):
            u, v, cm, sn, planes = pyorc_amd.piv_pairs(fr, ws, ov, thr, return_planes=True)
            n_rows, n_cols = u.shape[1:]
            _, _, corr = po.cross_corr(fr, ws, ov, signal_threshold=thr)
            uo, vo = po.u_v_displacement(corr, n_rows, n_cols)
            assert np.array_equal(np.isnan(planes), np.isnan(corr)), "NaN planes differ"
            assert np.nanmax(np.abs(planes - corr), initial=0.0) < 5e-6
            # windows whose arg-max is unique under float32 noise decide the border / NaN pattern
            flat = np.sort(np.nan_to_num(corr.reshape(corr.shape[0], corr.shape[1], -1), nan=0.0), axis=-1)
            uniq = ((flat[..., -1] - flat[..., -2]) > 1e-5 * np.maximum(flat[..., -1], 1e-12)).reshape(uo.shape) | np.isnan(uo)
            assert np.array_equal(np.isnan(u)[uniq], np.isnan(uo)[uniq])
            both = uniq & ~np.isnan(uo) & (np.abs(u - uo) < 0.5) & (np.abs(v - vo) < 0.5)
            assert both.sum() >= 0.5 * (uniq & ~np.isnan(uo)).sum()
            assert np.percentile(np.abs(u - uo)[both], 95) < 2e-3 and np.percentile(np.abs(v - vo)[both], 95) < 2e-3
            if opt == "border_peak":
                with po.semantics(border_peak=0):
                    u0, _ = po.u_v_displacement(corr, n_rows, n_cols)
                edge = np.isnan(u0) & ~np.isnan(corr[:, :, 0, 0]).reshape(u0.shape) & uniq
                assert edge.any(), "test input has no border peak"
                assert np.array_equal(u[edge], uo[edge].astype(np.float32)) and np.array_equal(v[edge], vo[edge].astype(np.float32))
                u2, v2 = pyorc_amd.u_v_displacement(planes, n_rows, n_cols)        # the plane-volume entry point
                assert np.array_equal(u2[edge], u[edge]) and np.array_equal(v2[edge], v[edge])
            else:
                with po.semantics(**{opt: 0}):
                    _, _, corr_default = po.cross_corr(fr, ws, ov, signal_threshold=thr)
                assert not np.array_equal(np.isnan(corr), np.isnan(corr_default)), "the option changes nothing on this input"
                ens = P.Ensemble((H, Wd), ws, ov)                                  # ensemble mode honours it too
                cme, _ = ens.accumulate(fr, 0.0, 0.0, thr)
                ens.close()
                dropped = np.isnan(corr[:, :, 0, 0])
                assert np.all(cme[dropped] == 0.0) and np.all(cme[~dropped & (cm.reshape(cme.shape) > 1e-6)] > 0.0)
    finally:
        pyorc_amd.set_option(opt, 0)


@pytest.mark.parametrize("ws", [(32, 32), (64, 64), (24, 24), (16, 16), (9, 9), (27, 27), (24, 16), (96, 96)])
@pytest.mark.parametrize("opt,val", [("v_sign", 1), ("std_ddof", 1), ("norm_clip", 0)])
def test_unpinned_semantics_switches_round3(gpu, opt, val, ws):
    """The four readings that became switches in round 3: v negated inside the engine, sample instead of
    population standard deviation, no clip of the normalised windows -- same names and values in `po.semantics` and in
    `lspiv_set_option`, every kernel family, per-timestep + plane volume + ensemble finish; all windows gated at 1e-4
    (the rescue pass honours the options).  `round_odd` is host-side: test_host.py."""
    import pyorc_amd
    import pyorc_amd.piv as P
    from pyorc_amd import _lib

    H, Wd = 3 * ws[0] + 5, 4 * ws[1] + 3
    fr = particle_stack(4, H, Wd, seed=70 + ws[0], density=0.05)
    ov = (ws[0] // 2, ws[1] // 2)
    base = pyorc_amd.piv_pairs(fr, ws, ov)
    default = 1 if opt == "norm_clip" else 0
    pyorc_amd.set_option(opt, val)
    try:
        if opt == "norm_clip":
            assert _lib.load().lspiv_kernel_kind(*ws) in (3, 9, 10)          # block-per-window kernels serve this reading
        with po.semantics(**{opt: val}):
            u, v, cm, sn, planes = pyorc_amd.piv_pairs(fr, ws, ov, return_planes=True)
            n_rows, n_cols = u.shape[1:]
            _, _, corr = po.cross_corr(fr, ws, ov)
            uo, vo, cmo, sno = po.get_uv_timestep(fr, n_cols, n_rows, ws, ov)
            assert np.nanmax(np.abs(planes - corr), initial=0.0) < 5e-6
            flat = np.sort(corr.reshape(corr.shape[0], corr.shape[1], -1), axis=-1)
            ok = ((flat[..., -1] - flat[..., -2]) > 1e-12 * np.maximum(flat[..., -1], 1e-300)).reshape(uo.shape)   # no exact tie
            assert ok.mean() > 0.9 and np.array_equal(np.isnan(u)[ok], np.isnan(uo)[ok])
            assert rel_err(cm, cmo.astype(np.float64)) <= TOL and rel_err(sn, sno.astype(np.float64)) <= TOL
            assert rel_err(u[ok], uo[ok].astype(np.float64)) <= TOL and rel_err(v[ok], vo[ok].astype(np.float64)) <= TOL
            u2, v2 = pyorc_amd.u_v_displacement(planes, n_rows, n_cols)        # the plane-volume entry point (float32 planes)
            both = ~np.isnan(u2) & ~np.isnan(uo) & (np.abs(u2 - uo) < 0.5) & (np.abs(v2 - vo) < 0.5)
            assert both.mean() > 0.5 and np.percentile(np.abs(v2 - vo)[both], 95) < 2e-3
        if opt == "v_sign":
            assert np.array_equal(v, -base[1], equal_nan=True) and np.array_equal(u, base[0], equal_nan=True)
            if ws[0] == ws[1] and ws[0] <= 64:                                 # ensemble finish honours it too
                for sign in (1, 0):
                    pyorc_amd.set_option("v_sign", sign)
                    ens = P.Ensemble((H, Wd), ws, ov)
                    ens.accumulate(fr, 0.0, 0.0)
                    ue, ve, _ = ens.finish(0.0, 1)
                    ens.close()
                    if sign:
                        first = (ue, ve)
                    else:
                        assert np.array_equal(first[0], ue, equal_nan=True) and np.array_equal(first[1], -ve, equal_nan=True)
        elif opt == "std_ddof":
            n = ws[0] * ws[1]
            assert np.allclose(cm, base[2] * (n - 1) / n, rtol=2e-6, equal_nan=True)
        else:
            assert not np.allclose(cm, base[2], rtol=1e-3, equal_nan=True), "the option changes nothing on this input"
    finally:
        pyorc_amd.set_option(opt, default)


# ------------------------------------------------------------------ get_ffpiv / get_piv -------------
def test_get_ffpiv_timestep_chunking_is_bit_identical(gpu):
    """get_ffpiv with the DEFAULT (time-walking) kernels returns the same bits for every chunk size -- the reference
    computes every window independently, so its result cannot depend on the chunking either
    (pyorc/velocimetry/ffpiv.py:140,399-442).  88 frames: three full 25-pair segments and a 12-pair tail."""
    from pyorc_amd import frames as F

    fr = particle_stack(88, 96, 128, seed=31)
    t = np.cumsum(np.r_[0.0, np.full(87, 1 / 30) + np.arange(87) * 1e-4])
    whole = F.get_piv(fr, 32, time=t, resolution=0.01)
    assert set(whole) == {"s2n", "corr", "v_x", "v_y"} and whole["v_x"].shape == (87, 5, 7)
    assert whole["v_x"].dtype == np.float32 and np.array_equal(whole.coords["time"], t[1:])
    assert np.array_equal(whole.coords["x"], np.arange(128)[16::16][:7])
    for cs in (2, 3, 5, 8, 25, 26, 49, 50, 60, 87, 88, 200):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            part = F.get_piv(fr, 32, time=t, resolution=0.01, chunksize=cs)
        for k in whole:
            assert np.array_equal(whole[k], part[k], equal_nan=True), (cs, k)
        assert np.array_equal(whole.coords["time"], part.coords["time"])
    ref = po.get_ffpiv(fr[:12], np.diff(t[:12]), (32, 32), (16, 16), 0.01, 0.01)
    for k in ("corr", "s2n"):
        assert rel_err(whole[k][:11], ref[k].astype(np.float64)) <= TOL
    floor = 0.05 * 0.01 * 30  # 0.05 px in m/s
    for k in ("v_x", "v_y"):
        assert rel_err(whole[k][:11], ref[k].astype(np.float64), floor=floor) <= TOL


@pytest.mark.parametrize("ws,dtype", [(32, np.uint8), (64, np.float32), (24, np.uint8), (16, np.float64), (10, np.uint8)])
def test_chunks_cut_on_anchors_reproduce_the_whole_stack(gpu, ws, dtype):
    """lspiv_piv_pairs_at: chunks that start on multiples of lspiv_chunk_alignment reproduce the one-call result bit for
    bit (time chunks, multi-GPU time blocks); a chunk that starts elsewhere is still right (oracle gate) and differs from
    the whole-stack run in its first, partial segment only -- from the next anchor on the bits are equal again."""
    import pyorc_amd
    from pyorc_amd import shard, window

    W = (ws, ws)
    ov = (ws // 2, ws // 2)
    fr = particle_stack(64, 2 * ws + 11, 3 * ws + 2, seed=77 + ws, density=0.05)
    fr = fr if dtype == np.uint8 else fr.astype(dtype) - 11.5
    A = window.chunk_alignment(W, fr.shape[1:], ov)          # the grid's own anchor length (a small grid: the family's base)
    assert A == 25 and window.chunk_alignment(W) == 75         # without a grid: the any-grid alignment (ABI 5)
    whole = np.stack(pyorc_amd.piv_pairs(fr, W, ov))
    for bounds in ([0, 25, 63], [0, 50, 63], [0, 25, 50, 63]):
        parts = [np.stack(pyorc_amd.piv_pairs(fr[a:b + 1], W, ov, pair_offset=a)) for a, b in zip(bounds, bounds[1:])]
        assert np.array_equal(np.concatenate(parts, axis=1), whole, equal_nan=True), bounds
    for world in (2, 3):                                               # the blocks pyorc_amd.shard hands to the ranks
        blocks = [shard.pair_block(63, r, world, A) for r in range(world)]
        parts = [np.stack(pyorc_amd.piv_pairs(fr[a:b + 1], W, ov, pair_offset=a)) for a, b in blocks if b > a]
        assert np.array_equal(np.concatenate(parts, axis=1), whole, equal_nan=True), world
    off = np.stack(pyorc_amd.piv_pairs(fr[10:], W, ov, pair_offset=10))          # off-anchor start
    assert np.array_equal(off[:, 15:], whole[:, 25:], equal_nan=True)             # pairs 25.. : identical again
    assert_same_to_rounding(off[:, :15], whole[:, 10:25])
    noff = np.stack(pyorc_amd.piv_pairs(fr[10:], W, ov))                          # same chunk, offset not given
    assert_same_to_rounding(noff, whole[:, 10:])


def assert_same_to_rounding(got, ref, corr_tol=1e-5, uv_tol=1e-4):
    """Two runs of the same frame pairs through DIFFERENT transform-sharing patterns (time-walking vs per-pair kernel, or
    an off-anchor chunk vs the whole stack) agree to float32 rounding: same NaN mask up to arg-max ties on a plane
    border, corr / s2n to 1e-5; displacements, wherever both runs picked the same peak: 99.9 % within 1e-4 of
    max(|ref|, 0.05 px) -- the rest are the ill-conditioned sub-pixel fits (flat ridges, empty neighbours) that amplify
    float32 rounding for any implementation (with the rescue pass on, both runs re-evaluate those in float64)."""
    u, v, c, s = (np.asarray(a, dtype=np.float64) for a in got)
    uo, vo, co, so = (np.asarray(a, dtype=np.float64) for a in ref)
    assert u.shape == uo.shape
    assert np.array_equal(np.isnan(c), np.isnan(co)) and np.array_equal(np.isnan(s), np.isnan(so))
    assert (np.isnan(u) != np.isnan(uo)).mean() < 1e-3
    with np.errstate(all="ignore"):
        assert np.nanmax(np.abs(c - co) / np.maximum(np.abs(co), 0.05), initial=0.0) <= corr_tol
        assert np.nanmax(np.abs(s - so) / np.maximum(np.abs(so), 0.05), initial=0.0) <= corr_tol
        same = (np.abs(u - uo) < 0.5) & (np.abs(v - vo) < 0.5)
        assert same[~np.isnan(u) & ~np.isnan(uo)].mean() > 0.999
        for g, r in ((u, uo), (v, vo)):
            e = (np.abs(g - r) / np.maximum(np.abs(r), 0.05))[same]
            e = e[~np.isnan(e)]
            assert e.size == 0 or (np.percentile(e, 99.9) <= uv_tol and e.max() <= 0.2)


def test_ensemble_chunking_is_bit_identical(gpu):
    """Ensemble mode: chunked accumulation with boundaries on the anchors leaves the same corr_sum / corr_count bits in
    HBM as one call over the whole stack (anchored segments, partial sums merged in segment order)."""
    import pyorc_amd.piv as P

    fr = particle_stack(64, 96, 128, seed=91)
    states = []
    for bounds in ([0, 63], [0, 25, 63], [0, 50, 63], [0, 25, 50, 63]):
        ens = P.Ensemble((96, 128), (32, 32), (16, 16))
        cms = [ens.accumulate(fr[a:b + 1], 0.1, 1.5)[0] for a, b in zip(bounds, bounds[1:])]
        states.append((np.concatenate(cms), *ens.export_state()))
        ens.close()
    for st in states[1:]:
        for a, b in zip(states[0], st):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("kw", [dict(corr_min=0.0, s2n_min=0.0, count_min=0.0), dict(), dict(corr_min=0.5, s2n_min=4.0),
                                dict(signal_threshold=0.3, count_min=0.5)])
def test_get_ffpiv_ensemble_vs_oracle(gpu, kw):
    from pyorc_amd import frames as F

    fr = particle_stack(9, 128, 160, seed=33)
    t = np.arange(9) / 25.0
    for cs in (None, 3):
        got = F.get_piv(fr, 32, time=t, resolution=0.02, ensemble_corr=True, chunksize=cs, **kw)
        ref = po.get_ffpiv(fr, np.diff(t), (32, 32), (16, 16), 0.02, 0.02, ensemble_corr=True, chunksize=cs, **kw)
        assert got["v_x"].shape == (1, 7, 9)
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), (k, cs)
        assert rel_err(got["corr"], ref["corr"]) <= TOL and rel_err(got["s2n"], ref["s2n"]) <= TOL
        floor = 0.05 * 0.02 * 25
        assert rel_err(got["v_x"], ref["v_x"].astype(np.float64), floor=floor) <= TOL
        assert rel_err(got["v_y"], ref["v_y"].astype(np.float64), floor=floor) <= TOL
        assert np.array_equal(got.coords["time"], t[ref["pair_index"]])


@pytest.mark.parametrize("n,T", [(24, 6), (10, 7), (16, 5), (26, 4), (48, 4), (20, 5), (12, 6), (40, 4), (36, 3), (28, 5), (34, 3), (6, 5), (18, 4), (30, 3), (-10, 7), (-26, 4), (-1034, 3), (62, 3), (42, 4), (8, 6),
                                 (64, 5), (64, 29), (48, 28), (32, 28)])   # long ones: several anchored segments, partial sums merged
def test_ensemble_other_window_size(gpu, monkeypatch, n, T):
    """Ensemble mode of the 16-point and the prime-factor FFT kernels (6 ... 48), of the direct kernel (34 with both switches) and, with
    LSPIV_NO_PFA=1 (negative n), of the embedded kernels (10: 32-point variant, two pairs per iteration incl. an odd pair
    count; 26: 64-point variant) that serve the odd sizes."""
    from pyorc_amd import _lib, frames as F

    if n < 0:
        n = -n
        monkeypatch.setenv("LSPIV_NO_PFA", "1")
        if n > 1000:                                   # and past the embedded kernels: the direct spatial kernel
            n -= 1000
            monkeypatch.setenv("LSPIV_NO_EMBED", "1")
        assert _lib.load().lspiv_kernel_kind(n, n) in ((4, 5) if n < 1000 and "LSPIV_NO_EMBED" not in os.environ else (3,))

    fr = particle_stack(T, 4 * n, 4 * n, seed=35 + n, density=0.05)
    got = F.get_piv(fr, n, ensemble_corr=True, corr_min=0.1, s2n_min=1.5)
    ref = po.get_ffpiv(fr, np.ones(T - 1), (n, n), (n // 2, n // 2), 1.0, 1.0, ensemble_corr=True, corr_min=0.1, s2n_min=1.5)
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), k
    assert rel_err(got["corr"], ref["corr"].astype(np.float64)) <= TOL and rel_err(got["s2n"], ref["s2n"].astype(np.float64)) <= TOL
    assert rel_err(got["v_x"], ref["v_x"].astype(np.float64)) <= TOL and rel_err(got["v_y"], ref["v_y"].astype(np.float64)) <= TOL


@pytest.mark.parametrize("n,ov,dtype,H,W,T", [(64, 48, np.uint8, 200, 640, 5), (64, 48, np.uint8, 160, 700, 28), (32, 16, np.float32, 100, 480, 28),
                                              (32, 16, np.float64, 90, 520, 5), (64, 32, np.float32, 200, 1300, 4)])
def test_ensemble_and_per_timestep_on_grids_wider_than_a_job_strip(gpu, n, ov, dtype, H, W, T):
    """The walking kernels run these shapes' jobs in column strips (strip_order: 32 windows at 64 x 64, 24 for 32 x 32 windows of
    float frames); the grids here have more columns than a strip, which the small frames of the other tests never have.  (Round 4
    shipped the ensemble kernel with its partial-sum slot indexed by the permuted job for a while: wrong sums from column 32 on,
    caught by tests/test_gpu_strip_order.py.)"""
    from pyorc_amd import frames as F

    fr = particle_stack(T, H, W, seed=91 + n + T, density=0.04)
    if dtype != np.uint8:
        fr = fr.astype(dtype) * 0.5 - 7.0
    for ens in (True, False):
        kw = dict(corr_min=0.1, s2n_min=1.5) if ens else {}
        got = F.get_piv(fr, n, overlap=(ov, ov), ensemble_corr=ens, **kw)
        ref = po.get_ffpiv(fr, np.ones(T - 1), (n, n), (ov, ov), 1.0, 1.0, ensemble_corr=ens, **kw)
        assert got["v_x"].shape[-1] > 24 + 8 * (n == 64)
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), (k, ens)
            assert rel_err(got[k], np.asarray(ref[k], dtype=np.float64)) <= TOL, (k, ens)


def test_pipelined_upload_equals_single_batch(gpu, monkeypatch):
    """Host entry point: staging the stack in sub-batches of frames (two pinned slots, compute overlapped with
    the next DMA; launches cut on the segment anchors) must equal one batch bit for bit with the default kernels, for
    every batch geometry including 1 frame per batch."""
    import pyorc_amd

    fr = particle_stack(58, 96, 128, seed=41)
    monkeypatch.setenv("LSPIV_STAGE_BYTES", str(1 << 30))
    ref = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16), return_planes=True)
    for stage in (1, 96 * 128 * 2 + 5, 96 * 128 * 5, 96 * 128 * 11, 96 * 128 * 26, 96 * 128 * 40):
        monkeypatch.setenv("LSPIV_STAGE_BYTES", str(stage))
        got = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16), return_planes=True)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b, equal_nan=True), stage
        got = pyorc_amd.piv_pairs(fr.astype(np.float64), (24, 24), (12, 12))
        monkeypatch.setenv("LSPIV_STAGE_BYTES", str(1 << 30))
        ref2 = pyorc_amd.piv_pairs(fr.astype(np.float64), (24, 24), (12, 12))
        for a, b in zip(ref2, got):
            assert np.array_equal(a, b, equal_nan=True), stage


def test_ensemble_state_export_import_merges_time_blocks(gpu):
    """Multi-GPU ensemble = per-rank accumulate + summed state: two half-stack ensembles merged through
    export_state / import_state must equal one ensemble over the whole stack (float32 fits: the rescue of the final fit is
    switched off here, the staged finish that keeps it across handles is tested below)."""
    import pyorc_amd.piv as P
    from pyorc_amd import _lib

    fr = particle_stack(11, 96, 128, seed=51)
    kw = dict(corr_min=0.1, s2n_min=1.5)
    _lib.set_option("rescue", 0)
    try:
        whole = P.Ensemble(fr.shape[1:], (32, 32), (16, 16))
        cm_w, sn_w = whole.accumulate(fr, **kw)
        a = P.Ensemble(fr.shape[1:], (32, 32), (16, 16))
        b = P.Ensemble(fr.shape[1:], (32, 32), (16, 16))
        cm_a, _ = a.accumulate(fr[:6], **kw)      # pairs 0..4
        cm_b, _ = b.accumulate(fr[5:], **kw)      # pairs 5..9 (one halo frame)
        # a plane shares its inverse FFT with the neighbouring pair OF ITS CHUNK: same values up to float32 rounding
        assert np.allclose(np.concatenate([cm_a, cm_b]), cm_w, rtol=2e-6, atol=1e-7)
        sa, ka = a.export_state()
        sb, kb = b.export_state()
        sw, kw_ = whole.export_state()
        assert np.array_equal(ka + kb, kw_) and np.abs((sa + sb) - sw).max() < 1e-5
        a.import_state(sb, kb, add=True)
        u1, v1, c1 = a.finish(0.2, 1)
        u0, v0, c0 = whole.finish(0.2, 1)
        assert np.array_equal(c1, c0) and np.array_equal(np.isnan(u1), np.isnan(u0))
        assert np.nanmax(np.abs(u1 - u0)) < 1e-4 and np.nanmax(np.abs(v1 - v0)) < 1e-4
        b.import_state(sw, kw_)                   # replace
        u2, v2, _ = b.finish(0.2, 1)
        assert np.array_equal(u2, u0, equal_nan=True) and np.array_equal(v2, v0, equal_nan=True)
        with pytest.raises(ValueError):
            b.import_state(sw[:3], kw_)
        for e in (whole, a, b):
            e.close()
    finally:
        _lib.set_option("rescue", 1)


def test_ensemble_staged_finish_over_two_handles(gpu):
    """The multi-GPU finish (include/lspiv.h: lspiv_ensemble_flag / _partials / _finish_partials): two handles hold one half of
    the pairs each and, after the exchange, the same total state; each contributes the float64 sums over ITS frames for the same
    sorted list of flagged windows; with the summed partials both arrive at the answer of one handle over all frames."""
    import pyorc_amd.piv as P

    T, H, W = 9, 200, 264
    fr = _speckle_and_particles(T, H, W, 12)
    ws, ov = (32, 32), (16, 16)
    kw = dict(corr_min=0.1, s2n_min=1.5)
    whole = P.Ensemble((H, W), ws, ov)
    whole.accumulate(fr, **kw)
    u0, v0, _ = whole.finish(0.2, 1)
    st0 = whole.stats()
    assert st0["rescued"] > 0
    a, b = P.Ensemble((H, W), ws, ov), P.Ensemble((H, W), ws, ov)
    a.accumulate(fr[:5], **kw)
    b.accumulate(fr[4:], **kw)
    (sa, ka), (sb, kb) = a.export_state(), b.export_state()
    for e in (a, b):
        e.import_state(sa + sb, ka + kb)
    na, nb = a.flag(0.2, 1), b.flag(0.2, 1)
    assert na == nb == st0["flagged"] > 0
    (pa, oka), (pb, okb) = a.partials(), b.partials()
    assert oka and okb and pa.shape == (na, 20) and np.any(pa != 0) and np.any(pb != 0)
    ua, va, _ = a.finish_partials(pa + pb)
    ub, vb, _ = b.finish_partials(pa + pb)
    assert np.array_equal(ua, ub, equal_nan=True) and np.array_equal(va, vb, equal_nan=True)
    assert np.array_equal(np.isnan(ua), np.isnan(u0))
    ref = po.get_ffpiv(fr, np.ones(T - 1), ws, ov, 1.0, 1.0, ensemble_corr=True, count_min=0.2, **kw)
    for got in ((ua, va), (u0, v0)):
        assert rel_err(got[0][0], ref["v_x"][0].astype(np.float64)) <= TOL and rel_err(got[1][0], ref["v_y"][0].astype(np.float64)) <= TOL
    uf, vf, _ = a.finish(0.2, 1)                  # plain finish on a shared state: float32 fits, and it says so
    assert a.stats()["rescued"] == 0 and a.stats()["float32_kept"] == na
    with pytest.raises(ValueError):
        a.finish_partials(pa[:-1])
    for e in (whole, a, b):
        e.close()


# ------------------------------------------------------------------ errors --------------------------
def test_error_mapping(gpu):
    import pyorc_amd
    from pyorc_amd import _lib

    with pytest.raises(ValueError):
        pyorc_amd.piv_pairs(np.zeros((1, 64, 64), np.uint8))          # one frame: no pair
    with pytest.raises((ValueError, _lib.LspivError)):
        pyorc_amd.piv_pairs(np.zeros((2, 16, 16), np.uint8))          # frame smaller than window
    with pytest.raises(ValueError) as ei:                                # the reference's exception type for a bad window
        pyorc_amd.piv_pairs(np.zeros((2, 1100, 1100), np.uint8), (514, 514), (257, 257))
    assert ei.value.code == _lib.LSPIV_EUNSUPPORTED and isinstance(ei.value, _lib.LspivError)
    with pytest.raises(_lib.LspivError) as ei:
        pyorc_amd.piv_pairs(np.zeros((2, 64, 64), np.uint8), (32, 32), (32, 16))
    assert ei.value.code == _lib.LSPIV_EINVAL


@pytest.mark.gpu
def test_ensemble_accumulate_dev_equals_host_variant(gpu):
    """lspiv_ensemble_accumulate_dev on an HBM-resident chunk == lspiv_ensemble_accumulate on the same host chunk."""
    import ctypes as C

    from pyorc_amd import _lib, piv

    lib = _lib.load()
    fr = particle_stack(6, 128, 160, seed=4)
    n_win = 7 * 9
    host = piv.Ensemble((128, 160), (32, 32), (16, 16))
    cm, sn = host.accumulate(fr, 0.2, 3.0)
    dev = piv.Ensemble((128, 160), (32, 32), (16, 16))
    d_f, d_o = C.c_void_p(), C.c_void_p()
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_f), fr.nbytes))
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 8 * 5 * n_win))
    _lib.check(lib.lspiv_memcpy_h2d(d_f, _lib.ptr(fr), fr.nbytes))
    dev.accumulate_dev(d_f.value, np.uint8, 6, 0.2, 3.0, d_o.value)
    out = np.empty((2, 5, n_win), np.float32)
    _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(out), d_o, out.nbytes))
    assert np.array_equal(out[0], cm, equal_nan=True) and np.array_equal(out[1], sn, equal_nan=True)
    for a, b in zip(host.finish(0.2, 5), dev.finish(0.2, 5)):
        assert np.array_equal(a, b, equal_nan=True)
    host.close(); dev.close(); lib.lspiv_dev_free(d_f); lib.lspiv_dev_free(d_o)


@pytest.mark.parametrize("n", [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 18, 19, 20, 21, 22, 24, 25, 26, 27, 28, 30, 31, 33, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62])
def test_embedded_windows_every_size(gpu, n):
    """Square windows 4..8, 9..15 and 21..31 run inside the 16- / 32- / 64-point FFT kernels (zero-padded a, periodic b: exact
    circular correlation in the top-left corner), 17..19 in the direct kernel; every even size has FFT kernels of its own (16 and the P x 2^m sizes).  Every size, three dtypes, threshold, constant / empty regions, planes."""
    from pyorc_amd import _lib

    pfa = [m for m in range(6, 64, 2) if m not in (8, 16, 32)]     # own FFT kernels (prime-factor P x 2^m transforms)
    assert _lib.load().lspiv_kernel_kind(n, n) == (8 if n in pfa else 6 if n in (8, 16) else 7 if n <= 8 else 4 if n < 16 else 3 if n <= 20 or n > 32 else 5)
    fr = particle_stack(4, 3 * n + 5, 4 * n + 3, seed=100 + n, density=0.06)
    ov = (n // 2, n // 3)
    # 16-sample windows inside a 1024-point transform: the periodic copy of b carries 64x the window's energy, which
    # costs the smallest sizes half a digit of float32 headroom on the planes (gate on corr / u / v unchanged: 1e-4)
    small = dict(plane_tol=2e-6 if n >= 8 else 5e-6)
    check_against_oracle(fr, (n, n), ov, **small)
    check_against_oracle(fr.astype(np.float32) * 0.37 - 11.0, (n, n), ov, thr=0.3, **small)
    f64 = fr.astype(np.float64) * 2.1 - 40.0
    f64[1] = 0.0                                       # an empty frame: two dead pairs, exact zeros
    f64[:, : n + 2, : n + 2] = -2.5                    # a constant corner: zero variance for any n
    check_against_oracle(f64, (n, n), ov, plane_tol=small["plane_tol"])


def test_embedded_and_direct_kernels_agree(gpu):
    """24 x 24 three ways: its own FFT kernel (default), embedded in the 64-point kernel (LSPIV_NO_PFA=1) and the direct
    spatial kernel (LSPIV_NO_PFA=1 LSPIV_NO_EMBED=1): same answers."""
    import subprocess
    import sys

    code = ("import numpy as np, pyorc_amd; from pyorc_amd.synth import particle_stack; "
            "fr = particle_stack(3, 90, 120, seed=8, density=0.05); "
            "np.save(sys.argv[1], np.stack(pyorc_amd.piv_pairs(fr, (24, 24), (12, 12)) + pyorc_amd.piv_pairs(fr, (10, 10), (5, 5))[:0]))")
    outs = []
    for env_extra in ({"LSPIV_NO_PFA": "1"}, {"LSPIV_NO_PFA": "1", "LSPIV_NO_EMBED": "1"}, {}):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"embed_ab_{len(outs)}.npy")
        subprocess.run([sys.executable, "-c", "import sys; " + code, path], check=True, env={**os.environ, **env_extra},
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        outs.append(np.load(path))
    cond = c_oracle.piv_pairs(particle_stack(3, 90, 120, seed=8, density=0.05), (24, 24), (12, 12), return_cond=True)[-1]
    ok = ~c_oracle.exact_tie(cond, c_oracle.piv_pairs(particle_stack(3, 90, 120, seed=8, density=0.05), (24, 24), (12, 12))[2])
    a = outs[0]                                                        # 64-point embedded kernel
    for b in outs[1:]:                                                 # direct kernel; the 24-point FFT kernel (default)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert rel_err(a[2:], b[2:].astype(np.float64)) <= 1e-5          # corr_max, s2n
        assert ok.mean() > 0.99 and rel_err(a[0][ok], b[0][ok].astype(np.float64)) <= TOL and rel_err(a[1][ok], b[1][ok].astype(np.float64)) <= TOL


@pytest.mark.parametrize("seg,P,ws", [("1", 3, 32), ("1", 4, 32), ("3", 10, 32), ("5", 11, 32), ("7", 23, 32), ("31", 40, 32),
                                      ("2", 9, 32), ("4", 9, 32), ("3", 7, 64), ("1", 5, 64), ("1", 6, 16), ("5", 12, 16),
                                      ("0", 4, 16), ("1", 6, 24), ("3", 8, 24), ("0", 4, 24), ("5", 11, 12), ("0", 3, 12),
                                      ("1", 5, 48), ("0", 3, 48), ("1", 7, 20), ("3", 5, 20), ("0", 3, 20), ("1", 4, 40), ("0", 3, 40), ("1", 6, 28), ("3", 4, 56), ("1", 4, 60), ("0", 3, 36), ("1", 7, 10), ("3", 6, 18), ("0", 3, 22), ("1", 4, 30), ("5", 6, 6), ("0", 3, 14), ("1", 3, 26), ("1", 4, 34), ("3", 3, 50), ("0", 3, 62), ("1", 6, 8), ("0", 3, 8)])
def test_walking_kernel_segments_vs_oracle(gpu, monkeypatch, seg, P, ws):
    """The time-walking kernel under every segment geometry (odd / even segment lengths, a last segment of one pair,
    an odd frame count) against the oracle: planes, thresholds, an empty frame and a constant corner in the stack."""
    monkeypatch.setenv("LSPIV_WALK", seg)
    fr = particle_stack(P + 1, 2 * ws + 9, 3 * ws + 5, seed=7 * P + ws, density=0.05)
    ov = (ws // 2, ws // 2)
    check_against_oracle(fr, (ws, ws), ov)
    f32 = fr.astype(np.float32) * 0.7 - 20.0
    f32[P // 2] = 3.0                                    # a constant frame: both pairs that touch it are dead
    f32[:, : ws + 4, : ws + 4] = -1.25                   # a constant corner in every frame
    check_against_oracle(f32, (ws, ws), ov, thr=0.25)
    if ws == 32:
        check_against_oracle(fr.astype(np.float64) - 3.0, (ws, ws), (20, 7))
    monkeypatch.setenv("LSPIV_WALK", "0")                # and the per-pair kernel agrees to rounding
    import pyorc_amd

    ref = pyorc_amd.piv_pairs(fr, (ws, ws), ov)
    monkeypatch.setenv("LSPIV_WALK", seg)
    # windows below 16 x 16: ~100-200 vectors, so the 99.9th percentile is the single worst-conditioned sub-pixel fit
    # (the oracle comparisons above grade those by their condition number)
    assert_same_to_rounding(pyorc_amd.piv_pairs(fr, (ws, ws), ov), ref)


def test_float64_frames_are_narrowed_while_staged(gpu):
    """Host entry points convert float64 stacks to float32 in the staging threads; the kernels would do the same
    conversion on load, so results equal those of a float32 copy of the stack bit for bit (incl. non-representable
    values and several staging sub-batches), per-timestep and ensemble mode."""
    import pyorc_amd
    import pyorc_amd.piv as P

    rng = np.random.default_rng(3)
    fr = particle_stack(9, 100, 140, seed=77).astype(np.float64) * 1.000000123 + rng.random((9, 100, 140)) * 1e-3 - 17.3
    os.environ["LSPIV_STAGE_BYTES"] = str(100 * 140 * 4 * 2 + 7)       # two frames per staging slot
    try:
        a = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16), 0.1)
        b = pyorc_amd.piv_pairs(fr.astype(np.float32), (32, 32), (16, 16), 0.1)
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)
        ea, eb = P.Ensemble(fr.shape[1:], (32, 32), (16, 16)), P.Ensemble(fr.shape[1:], (32, 32), (16, 16))
        ca, sa = ea.accumulate(fr, 0.1, 1.5)
        cb, sb = eb.accumulate(fr.astype(np.float32), 0.1, 1.5)
        assert np.array_equal(ca, cb) and np.array_equal(sa, sb)
        for x, y in zip(ea.finish(0.2, 1), eb.finish(0.2, 1)):
            assert np.array_equal(x, y, equal_nan=True)
        ea.close(); eb.close()
    finally:
        os.environ.pop("LSPIV_STAGE_BYTES", None)


def test_pinned_host_stack_is_used_in_place(gpu):
    """pyorc_amd.pinned_empty: a stack in page-locked host memory takes the no-staging path of lspiv_piv_pairs (same
    sub-batches, same kernels) and must give the results of a pageable copy bit for bit."""
    import pyorc_amd

    fr = particle_stack(10, 96, 128, seed=9)
    pin = pyorc_amd.pinned_empty(fr.shape, np.uint8)
    pin[...] = fr
    os.environ["LSPIV_STAGE_BYTES"] = str(96 * 128 * 3)
    try:
        a = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16))
        b = pyorc_amd.piv_pairs(pin, (32, 32), (16, 16))
    finally:
        os.environ.pop("LSPIV_STAGE_BYTES", None)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    view = pin[2:7]                      # a view keeps the pinned allocation alive
    del pin
    assert np.array_equal(view, fr[2:7])
    f32 = pyorc_amd.pinned_empty((4, 64, 64), np.float32)
    f32[...] = particle_stack(4, 64, 64, seed=2)
    for x, y in zip(pyorc_amd.piv_pairs(f32, (32, 32), (16, 16)), pyorc_amd.piv_pairs(np.array(f32), (32, 32), (16, 16))):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("n", [8, 16, 32, 64] + [m for m in range(6, 64, 2) if m not in (8, 16, 32)])
def test_register_ffts_match_numpy(gpu, n):
    """The kernels' own register transforms (fft_regs.h: radix-4/8 for the powers of two, Good-Thomas prime-factor
    splits with hand-written 3- / 5-point and generic odd-P butterflies for every other even length) against
    numpy.fft, forward and inverse, through the lspiv_debug_fft test hook: 64 random transforms, a delta and a constant."""
    from pyorc_amd import _lib

    rng = np.random.default_rng(n)
    z = (rng.standard_normal((66, n)) + 1j * rng.standard_normal((66, n))).astype(np.complex64)
    z[64] = 0; z[64, 1] = 1.0                        # delta at 1: the twiddle row itself
    z[65] = 2.5 - 1.0j                               # constant: only the DC bin
    out = np.empty_like(z)
    for inverse in (0, 1):
        _lib.check(gpu.lspiv_debug_fft(n, inverse, _lib.ptr(z), _lib.ptr(out), z.shape[0]))
        ref = np.fft.ifft(z.astype(np.complex128), axis=1) * n if inverse else np.fft.fft(z.astype(np.complex128), axis=1)
        scale = np.abs(ref).max(axis=1, keepdims=True)
        assert np.max(np.abs(out - ref) / scale) < 2e-6, (n, inverse)
    assert gpu.lspiv_debug_fft(7, 0, _lib.ptr(z), _lib.ptr(out), 1) == _lib.LSPIV_EUNSUPPORTED


# ------------------------------------------------------------------ float64 rescue pass (round 3) ---
def _rescue_stats(stream=None):
    from pyorc_amd import _lib

    st = (C.c_int64 * 5)()
    _lib.check(_lib.load().lspiv_rescue_stats(stream, st))
    return list(st)


def test_rescue_pass_touches_only_the_flagged_windows(gpu):
    """Rescue on vs off: the unflagged windows keep their bits (the PIV kernel is the same launch), the flagged ones move to
    the float64 answer; the counters say how many; `rescue = 0` reproduces round 2's float32 results (loosely gated)."""
    import pyorc_amd
    from pyorc_amd import _lib

    fr = particle_stack(30, 200, 260, seed=11, density=0.015)      # sparse seeding: plenty of ill-conditioned peaks
    ws, ov = (32, 32), (16, 16)
    on = pyorc_amd.piv_pairs(fr, ws, ov)
    st = _rescue_stats()
    n_tiles = on[0].size
    assert st[0] + st[1] > 0 and st[0] + st[1] < 0.2 * n_tiles
    _lib.set_option("rescue", 0)
    try:
        off = pyorc_amd.piv_pairs(fr, ws, ov)
    finally:
        _lib.set_option("rescue", 1)
    assert np.array_equal(on[2], off[2], equal_nan=True) and np.array_equal(on[3], off[3], equal_nan=True)   # corr, s2n untouched
    moved = ~((on[0] == off[0]) | (np.isnan(on[0]) & np.isnan(off[0]))) | ~((on[1] == off[1]) | (np.isnan(on[1]) & np.isnan(off[1])))
    assert 0 < moved.sum() <= st[0] + st[1]
    uo, vo, cmo, sno, cond = c_oracle.piv_pairs(fr, ws, ov, return_cond=True)
    ok = ~c_oracle.exact_tie(cond, cmo)
    assert rel_err(on[0][ok], uo[ok].astype(np.float64)) <= TOL and rel_err(on[1][ok], vo[ok].astype(np.float64)) <= TOL
    assert max(rel_err(off[0][ok], uo[ok].astype(np.float64)), rel_err(off[1][ok], vo[ok].astype(np.float64))) > TOL   # what the pass is for
    good = ok & c_oracle.well_posed(cond)
    assert rel_err(off[0][good], uo[good].astype(np.float64)) <= TOL                                       # round 2's gate still holds without it


def test_rescue_list_overflow_keeps_float32_results(gpu):
    """More flagged windows than the lists hold (a quarter of the launch): the excess keeps its float32 result, nothing
    crashes, the counters report the demand, and the next launch starts from clean counters."""
    import pyorc_amd

    rng = np.random.default_rng(3)
    T, H, W = 41, 528, 528                                          # 40 pairs x 32 x 32 windows = 40 960 > 4 x 4 096
    fr = np.zeros((T, H, W), np.uint8)
    ys, xs = np.meshgrid(np.arange(8, H, 16), np.arange(8, W, 16), indexing="ij")
    for t in range(T):                                              # one single-pixel speckle per window cell, drifting: every peak
        fr[t, (ys + t // 4) % H, (xs + t // 3) % W] = rng.integers(100, 255, ys.shape)   # sits on an exactly-zero neighbourhood
    u, v, cm, sn = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16))
    st = _rescue_stats()
    n_tiles = u.size
    assert n_tiles == 40 * 32 * 32 and st[0] + st[1] > n_tiles // 4           # demand above the capacity of the fit list
    assert np.isfinite(cm).all() and (np.isfinite(u) | np.isnan(u)).all()
    small = particle_stack(3, 96, 128, seed=5)
    a = pyorc_amd.piv_pairs(small, (32, 32), (16, 16))
    st2 = _rescue_stats()
    assert st2[0] + st2[1] <= a[0].size and st2[4] == st[4] + a[0].size        # counters were reset by the previous pass
    b = pyorc_amd.piv_pairs(small, (32, 32), (16, 16))
    assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))     # and the results are reproducible


def test_rescue_of_a_many_way_tie_in_a_tall_window(gpu):
    """the whole-plane ("amb") path of the rescue pass with more window rows than its block has threads (512 x 8:
    wy > 256).  Frames whose columns alternate give FOUR exactly equal maxima per plane (lags kx = 0, 2, 4, 6 of the true row
    shift): a 4-way near-tie in float32 -> an amb record -> the float64 plane -> its first maximum in row-major order, which sits
    in shifted column 0, on the border.  With the "integer peak" reading of border peaks the answer is exact: u = -4, v = +3."""
    import pyorc_amd
    from pyorc_amd import _lib

    rng = np.random.default_rng(5)
    H, W = 520, 24
    r = rng.integers(20, 200, H + 8).astype(np.float32)
    col = np.where(np.arange(W) % 2 == 0, 0.2, 1.0).astype(np.float32)
    fr = np.stack([r[t * 3: t * 3 + H, None] * col[None, :] for t in range(3)]).astype(np.float32)   # rows move by 3 per frame
    assert _lib.load().lspiv_kernel_kind(512, 8) == 9
    _lib.set_option("border_peak", 2)
    try:
        u, v, cm, sn = pyorc_amd.piv_pairs(fr, (512, 8), (0, 0))
        st = _rescue_stats()
    finally:
        _lib.set_option("border_peak", 0)
    assert u.shape == (2, 1, 3) and st[1] >= 1, st                 # whole-plane records were processed
    assert np.array_equal(u, np.full_like(u, -4.0)) and np.array_equal(np.abs(v), np.full_like(v, 3.0)), (u, v)
    assert np.all(cm > 0.5)


def _speckle_and_particles(T, H, W, seed):
    """Left part: one single-pixel speckle per 16 x 16 cell drifting one pixel per frame -- every correlation peak sits on
    exactly-zero neighbours, the worst case of the float32 fit; right part: an ordinary sparse particle image."""
    rng = np.random.default_rng(seed)
    fr = particle_stack(T, H, W, seed=seed, density=0.012)
    half = (W // 32) * 16
    fr[:, :, :half] = 0
    ys, xs = np.meshgrid(np.arange(8, H, 16), np.arange(8, half - 8, 16), indexing="ij")
    amp = rng.integers(100, 255, ys.shape)
    for t in range(T):
        fr[t, ys, xs + t] = amp
    return fr


def test_ensemble_rescue_of_the_final_fit(gpu, monkeypatch):
    """Round 4: lspiv_ensemble_finish re-evaluates the ill-conditioned fits of the MEAN planes in float64 from the retained frames
    (csrc/piv_rescue.hip, ens_*): every window within 1e-4 of the oracle; the host entry point keeps its upload buffers, the
    device entry point copies / borrows on request, and without frames (mode NONE, budget, imported state) the float32 fits stay."""
    import pyorc_amd.piv as P
    from pyorc_amd import DeviceFrames, _lib

    T, H, W = 5, 200, 264
    fr = _speckle_and_particles(T, H, W, 11)
    ws, ov = (32, 32), (16, 16)
    kw = dict(corr_min=0.1, s2n_min=1.5)
    ref = po.get_ffpiv(fr, np.ones(T - 1), ws, ov, 1.0, 1.0, ensemble_corr=True, count_min=0.0, **kw)
    uo, vo = ref["v_x"][0].astype(np.float64), ref["v_y"][0].astype(np.float64)

    def finish(ens):
        u, v, cnt = ens.finish(0.0, 1)
        st = ens.stats()
        ens.close()
        return u[0], v[0], st

    def check(u, v):
        assert np.array_equal(np.isnan(u), np.isnan(uo)) and np.array_equal(np.isnan(v), np.isnan(vo))
        return max(rel_err(u, uo), rel_err(v, vo))

    ens = P.Ensemble((H, W), ws, ov)                                 # host frames, one chunk
    ens.accumulate(fr, kw["corr_min"], kw["s2n_min"])
    u1, v1, st = finish(ens)
    assert st["flagged"] > 0 and st["rescued"] == st["flagged"] - st["float32_kept"] > 0 and st["retain_complete"] and st["chunks_kept"] == 1, st
    assert check(u1, v1) <= TOL
    ens = P.Ensemble((H, W), ws, ov)                                 # two chunks with their halo frame
    ens.accumulate(fr[:3], kw["corr_min"], kw["s2n_min"])
    ens.accumulate(fr[2:], kw["corr_min"], kw["s2n_min"])
    u2, v2, st2 = finish(ens)
    assert st2["chunks_kept"] == 2 and st2["rescued"] > 0 and check(u2, v2) <= TOL
    _lib.set_option("rescue", 0)                                     # what the pass is for
    try:
        ens = P.Ensemble((H, W), ws, ov)
        ens.accumulate(fr, kw["corr_min"], kw["s2n_min"])
        u0, v0, st0 = finish(ens)
    finally:
        _lib.set_option("rescue", 1)
    assert st0["flagged"] == 0 and st0["chunks_kept"] == 0 and check(u0, v0) > TOL
    d = DeviceFrames.from_host(fr)                                   # HBM-resident stack: borrowed, no copy
    ens = P.Ensemble((H, W), ws, ov)
    ens.accumulate(d, kw["corr_min"], kw["s2n_min"])
    u3, v3, st3 = finish(ens)
    assert np.array_equal(u3, u1, equal_nan=True) and np.array_equal(v3, v1, equal_nan=True) and st3["rescued"] == st["rescued"]
    d_cs = DeviceFrames.empty((2, T - 1, u1.size), np.float32)
    for mode, rescued in ((P.Ensemble.RETAIN_COPY, True), (P.Ensemble.RETAIN_NONE, False)):
        ens = P.Ensemble((H, W), ws, ov)
        ens.set_retain(mode)
        ens.accumulate_dev(d.ptr, d.dtype, T, kw["corr_min"], kw["s2n_min"], d_cs.ptr)
        u4, v4, st4 = finish(ens)
        if rescued:
            assert np.array_equal(u4, u1, equal_nan=True) and np.array_equal(v4, v1, equal_nan=True) and st4["bytes_kept"] >= fr.nbytes
        else:
            assert st4["flagged"] == st["flagged"] and st4["rescued"] == 0 and st4["float32_kept"] == st4["flagged"] and not st4["retain_complete"]
            assert np.array_equal(u4, u0, equal_nan=True) and np.array_equal(v4, v0, equal_nan=True)
    monkeypatch.setenv("LSPIV_ENSEMBLE_RETAIN_BYTES", "1000")        # budget too small for the chunk: float32 fits, nothing breaks
    ens = P.Ensemble((H, W), ws, ov)
    ens.accumulate(fr, kw["corr_min"], kw["s2n_min"])
    u5, v5, st5 = finish(ens)
    assert not st5["retain_complete"] and st5["rescued"] == 0 and np.array_equal(u5, u0, equal_nan=True)
    monkeypatch.delenv("LSPIV_ENSEMBLE_RETAIN_BYTES")
    ens = P.Ensemble((H, W), ws, ov)                                 # an imported state has no frames behind it
    ens.accumulate(fr, kw["corr_min"], kw["s2n_min"])
    s, k = ens.export_state()
    ens.import_state(s, k)
    u6, v6, st6 = finish(ens)
    assert st6["rescued"] == 0 and np.array_equal(u6, u0, equal_nan=True)


@pytest.mark.parametrize("ws,dtype,wide", [(64, np.uint8, 0), (64, np.float32, 0), (24, np.float64, 0), (9, np.uint8, 0), (40, np.float32, 0), (96, np.uint8, 0),
                                           (64, np.uint8, 1), (32, np.float32, 1), (32, np.uint8, 1)])   # wide: 39 columns, more than a job strip
def test_ensemble_rescue_other_kernels(gpu, ws, dtype, wide):
    """The same through get_piv for the other kernel families and sample types (64 x 64 float32 windows do not fit the rescue
    kernel's LDS slice and are read from L2; odd and > 64 px windows come from the embedded / DFT kernels)."""
    from pyorc_amd import frames as F

    fr = _speckle_and_particles(4, (3 if wide else 4) * ws + 8, (20 if wide else 6) * ws + 8, 100 + ws)
    if wide:
        fr = np.ascontiguousarray(fr[:, :, ::-1])   # the speckle half (where the fits get flagged) on the right: columns beyond the first strip
    if dtype != np.uint8:
        fr = fr.astype(dtype) * 0.5 - 3.0      # (the empty background stays exactly constant)
    ws_e = int(np.round(ws / 2.0) * 2)
    ov_e = int(round(ws) / 2)
    got = F.get_piv(fr, ws, ensemble_corr=True, corr_min=0.1, s2n_min=1.5, count_min=0.0)
    ref = po.get_ffpiv(fr, np.ones(3), (ws_e, ws_e), (ov_e, ov_e), 1.0, 1.0, ensemble_corr=True, corr_min=0.1, s2n_min=1.5, count_min=0.0)
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), k
    assert rel_err(got["v_x"], ref["v_x"].astype(np.float64)) <= TOL and rel_err(got["v_y"], ref["v_y"].astype(np.float64)) <= TOL


def test_two_host_threads_on_one_stream(gpu):
    """the PIV kernel and its rescue kernels share the stream's lists -- two host threads launching on the same
    (library) stream must not interleave them.  Different stacks and grid sizes from two threads, against the serial answers."""
    import threading

    import pyorc_amd
    from pyorc_amd import DeviceFrames

    stacks = [DeviceFrames.from_host(particle_stack(9, 200, 260, seed=61, density=0.012)),
              DeviceFrames.from_host(particle_stack(4, 96, 128, seed=62, density=0.012))]
    serial = [pyorc_amd.piv_pairs(s, (32, 32), (16, 16)) for s in stacks]
    errors = []

    def work(k):
        try:
            for _ in range(25):
                got = pyorc_amd.piv_pairs(stacks[k], (32, 32), (16, 16))
                for a, b in zip(serial[k], got):
                    if not np.array_equal(a, b, equal_nan=True):
                        errors.append(k)
                        return
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_float64_stack_on_a_large_dc_offset(gpu):
    """a float64 host stack whose texture (sigma ~ 1) rides on a DC offset of 1e4.  The reference normalises
    every window in float64; narrowing the frames to float32 as they are would keep the texture to 1e-3 only.  The staging threads
    take an integer estimate of each frame's offset off first (option "narrow_offset", csrc/host_stage.cpp) -- the per-window
    normalisation does not see it --, and the 1e-4 gate holds on every window, per time step and in ensemble mode."""
    import pyorc_amd
    from pyorc_amd import _lib, frames as F

    fr = particle_stack(5, 128, 160, seed=9, density=0.03).astype(np.float64) / 60.0 + 1.0e4
    ws, ov = (32, 32), (16, 16)
    check_against_oracle(fr, ws, ov, plane_tol=4e-6)
    got = F.get_piv(fr, 32, ensemble_corr=True, corr_min=0.1, s2n_min=1.5, count_min=0.0)
    ref = po.get_ffpiv(fr, np.ones(4), ws, ov, 1.0, 1.0, ensemble_corr=True, corr_min=0.1, s2n_min=1.5, count_min=0.0)
    assert np.array_equal(np.isnan(got["v_x"]), np.isnan(ref["v_x"]))
    assert rel_err(got["v_x"], ref["v_x"].astype(np.float64)) <= TOL and rel_err(got["v_y"], ref["v_y"].astype(np.float64)) <= TOL
    uo, vo, cmo, sno = c_oracle.piv_pairs(fr, ws, ov)
    _lib.set_option("narrow_offset", -1)                      # the plain conversion: what the option is for
    try:
        u, v, cm, sn = pyorc_amd.piv_pairs(fr, ws, ov)
    finally:
        _lib.set_option("narrow_offset", 1024)
    assert max(rel_err(u, uo.astype(np.float64)), rel_err(cm, cmo.astype(np.float64))) > TOL
    assert _lib.get_option("narrow_offset") == 1024


def test_rescue_totals_survive_a_regrow_and_streams_can_be_released(gpu):
    """the cumulative counters of lspiv_rescue_stats live in the lists' header; when a larger launch makes the lists grow
    the header moves along.  lspiv_stream_release drops the lists of a stream (the library's own: NULL); the next launch starts new ones."""
    import pyorc_amd
    from pyorc_amd import _lib

    lib = _lib.load()
    _lib.check(lib.lspiv_stream_release(None))                      # whatever earlier tests left on the library's stream
    small = particle_stack(3, 96, 128, seed=5)
    a = pyorc_amd.piv_pairs(small, (32, 32), (16, 16))
    st1 = _rescue_stats()
    assert st1[4] == a[0].size                                      # fresh lists: totals start at this launch
    big = particle_stack(40, 400, 528, seed=6, density=0.015)       # 39 x 24 x 32 windows: more than the first lists hold -> regrow
    b = pyorc_amd.piv_pairs(big, (32, 32), (16, 16))
    st2 = _rescue_stats()
    assert b[0].size // 4 > 4096 and st2[4] == st1[4] + b[0].size and st2[2] == st1[2] + st2[0] and st2[3] == st1[3] + st2[1]
    _lib.check(lib.lspiv_stream_release(None))
    assert _rescue_stats()[4] == 0                                  # no lists: nothing to report
    c = pyorc_amd.piv_pairs(small, (32, 32), (16, 16))
    assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, c)) and _rescue_stats()[4] == a[0].size
