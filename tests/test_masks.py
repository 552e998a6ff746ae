"""post-PIV masks (pyorc/api/mask.py) -- oracle self-checks on CPU, HIP parity on the GPU."""
import warnings

import numpy as np
import pytest

from oracle import mask_oracle as mo


def fields(T=12, R=9, C=11, seed=0, nan_frac=0.15):
    rng = np.random.default_rng(seed)
    f = np.empty((4, T, R, C), np.float32)
    f[0] = rng.normal(0.6, 0.5, (T, R, C))
    f[1] = rng.normal(-0.1, 0.3, (T, R, C))
    f[2] = rng.random((T, R, C))
    f[3] = rng.random((T, R, C)) * 30
    bad = rng.random((T, R, C)) < nan_frac
    f[:, bad] = np.nan
    f[:, :, 0, 0] = np.nan                      # an all-NaN cell
    f[:, :2, 1, 1] = 0.25                       # a constant (zero-variance) cell
    f[:, 2:, 1, 1] = np.nan
    f[:2, 0, 2, 2] = 0.0                        # zero speed
    return f


CASES = [
    ("minmax", dict(s_min=0.1, s_max=5.0), [0.1, 5.0]),
    ("minmax", dict(s_min=0.4, s_max=0.9), [0.4, 0.9]),
    ("count", dict(tolerance=0.33), [0.33]),
    ("count", dict(tolerance=0.9), [0.9]),
    ("corr", dict(tolerance=0.1), [0.1]),
    ("s2n", dict(tolerance=10), [10]),
    ("outliers", dict(tolerance=1.0, mode="or"), [1.0, 0]),
    ("outliers", dict(tolerance=0.7, mode="and"), [0.7, 1]),
    ("variance", dict(tolerance=5, mode="and"), [5, 1]),
    ("variance", dict(tolerance=1e-31, mode="or"), [1e-31, 0]),
    ("rolling", dict(wdw=5, tolerance=0.5), [5, 0.5]),
    ("rolling", dict(wdw=4, tolerance=0.8), [4, 0.8]),
    ("rolling", dict(wdw=1, tolerance=0.5), [1, 0.5]),
    ("window_nan", dict(tolerance=0.7, wdw=1), [0.7, -1, 1, -1, 1]),
    ("window_nan", dict(tolerance=0.5, wdw=2, wdw_x_min=-3, wdw_y_max=3), [0.5, -3, 2, -2, 3]),
    ("window_mean", dict(tolerance=0.7, wdw=1, mode="or"), [0.7, 0, -1, 1, -1, 1]),
    ("window_mean", dict(tolerance=0.3, wdw=2, mode="and"), [0.3, 1, -2, 2, -2, 2]),
]


def test_oracle_quirks_and_shapes():
    f = fields()
    assert len(mo._strides(1)) == 6 and (0, 1) not in mo._strides(1) and (1, -1) in mo._strides(1)   # y range excludes +wdw
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    s = mo._shift(a, 1, -1)                                             # out[r, c] = a[r + 1, c - 1]
    assert s[0, 1] == a[1, 0] and np.isnan(s[2]).all() and np.isnan(s[:, 0]).all()
    r = mo.rolling(f, wdw=5)
    assert not r[:2].any() and not r[-2:].any() and r[2:-2].any()      # incomplete windows mask everything
    assert not mo.rolling(f, wdw=4)[:2].any() and not mo.rolling(f, wdw=4)[-1].any() and mo.rolling(f, wdw=4)[-2].any()
    v = mo.variance(f)
    assert v.shape == f.shape[2:] and not v[0, 0] and v[3, 3]           # "std is not NaN" (np.maximum(mean, 1e30))
    assert mo.count(f).shape == f.shape[2:] and not mo.count(f)[0, 0]
    assert mo.time_mean(f).shape == (4, 1) + f.shape[2:]
    m = mo.minmax(f)
    g = mo.apply(f, m)
    assert np.isnan(g[:, ~m]).all() and np.array_equal(g[:, m], f[:, m])
    assert np.array_equal(mo.apply(f, mo.count(f))[:, :, mo.count(f)], f[:, :, mo.count(f)], equal_nan=True)
    rep = mo.window_replace(f, wdw=1, iter=2)
    assert np.isnan(rep).sum() < np.isnan(f).sum() and np.array_equal(rep[~np.isnan(f)], f[~np.isnan(f)])
    u = np.float32([[[1.5]], [[-2.0]]])
    vx, vy = mo.scale_velocity(u, u, 0.01, -0.01, [0.04, 0.05])
    assert vx.dtype == np.float32 and np.allclose(vx.ravel(), [0.375, -0.4]) and np.allclose(vy, -vx)


def test_mask_wrapper_assertions_without_device():
    from pyorc_amd.mask import Mask

    f = fields(T=1)
    ds = {k: f[i, 0] for i, k in enumerate(("v_x", "v_y", "corr", "s2n"))}
    with pytest.raises(AssertionError, match='This mask requires dimension "time"'):
        Mask(ds).variance()
    ds1 = {k: f[i] for i, k in enumerate(("v_x", "v_y", "corr", "s2n"))}
    with pytest.warns(UserWarning, match="This mask requires multiple timesteps"):
        m = Mask(ds1).variance()
    assert m.shape == f.shape[2:] and bool(m.all())
    with pytest.raises(AssertionError, match="not a valid velocimetry dataset"):
        Mask({**ds, "corr": f[2, 0, :3]}).corr()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,shape", [(0, (12, 9, 11)), (1, (2, 3, 70)), (2, (40, 17, 5)), (3, (400, 66, 119))])   # the last: a 1080p grid over 400 pairs
def test_gpu_masks_equal_oracle(gpu, seed, shape):
    from pyorc_amd import mask as pm

    f = fields(*shape, seed=seed)
    for name, kw, params in CASES:
        got = pm.run_mask(f, name, params)
        ref = getattr(mo, name)(f, **kw)
        assert got.shape == ref.shape and np.array_equal(got, ref), (name, kw, int((got != ref).sum()))
    got, ref = pm.run_mask(f, "angle", [0.5 * np.pi, 0.25 * np.pi]), mo.angle(f)
    with np.errstate(invalid="ignore"):
        edge = np.abs(np.abs(np.arctan2(f[0], f[1]) - np.float32(0.5 * np.pi)) - np.float32(0.25 * np.pi)) < 1e-5
    assert np.array_equal(got[~edge], ref[~edge]) and edge.sum() < max(3, 2e-5 * edge.size)   # atan2f is not correctly rounded: cells within 1e-5 rad of the tolerance
    assert np.array_equal(pm.time_mean(f), mo.time_mean(f), equal_nan=True)
    for m in (mo.minmax(f), mo.count(f)):
        assert np.array_equal(pm.apply_mask(f, m), mo.apply(f, m), equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [2, 7, 8, 9, 20, 64, 128, 129, 136, 300, 1000])
def test_gpu_time_reductions_on_a_single_cell_grid(gpu, T):
    """A window grid of ONE cell (R = C = 1): the time axis is the array's contiguous inner axis and numpy sums it PAIRWISE (below 8
    elements in order, up to 128 in eight strided partial sums, above that by halves) instead of sequentially as for every other
    grid -- found by tools/fuzz_rows.py seed 1108 in round 5 (time_mean differed in the last bit).  time_mean and the masks built on
    nanmean / nanstd over time, every length class of the pairwise scheme, with NaNs."""
    from pyorc_amd import mask as pm

    rng = np.random.default_rng(T)
    f = np.empty((4, T, 1, 1), np.float32)
    f[0] = rng.normal(0.6, 0.5, (T, 1, 1)); f[1] = rng.normal(-0.1, 0.3, (T, 1, 1)); f[2] = rng.random((T, 1, 1)); f[3] = rng.random((T, 1, 1)) * 30
    f[:, rng.random((T, 1, 1)) < 0.2] = np.nan
    if T > 2:
        f[:, 1] = np.float32(0.37)      # at least two finite samples
        f[:, 2] = np.float32(-0.21)
    assert np.array_equal(pm.time_mean(f), mo.time_mean(f), equal_nan=True)
    for name, kw, params in (("outliers", dict(tolerance=1.0, mode="or"), [1.0, 0]), ("outliers", dict(tolerance=0.7, mode="and"), [0.7, 1]),
                             ("variance", dict(tolerance=5, mode="and"), [5, 1]), ("variance", dict(tolerance=1e-31, mode="or"), [1e-31, 0]),
                             ("count", dict(tolerance=0.5), [0.5])):
        got, ref = pm.run_mask(f, name, params), getattr(mo, name)(f, **kw)
        assert got.shape == ref.shape and np.array_equal(got, ref), (name, kw)
    # and a grid of two cells keeps the sequential order (the reduction axis is no longer the inner one)
    g = np.concatenate([f, f[:, :, :, ::-1] * np.float32(1.5)], axis=3)
    assert np.array_equal(pm.time_mean(g), mo.time_mean(g), equal_nan=True)


@pytest.mark.gpu
def test_gpu_mask_methods_follow_the_reference_wrapper(gpu):
    from pyorc_amd.mask import Mask
    from pyorc_amd.velocimetry import PivResult

    f = fields(14, 10, 13, seed=5)
    mk = lambda: PivResult({k: f[i].copy() for i, k in enumerate(("v_x", "v_y", "corr", "s2n"))}, {})
    ds = mk()
    m1 = Mask(ds).angle()
    m2 = Mask(ds).window_mean(reduce_time=True)                               # (y, x) mask from the time mean
    assert m1.shape == f.shape[1:] and m2.shape == f.shape[2:]
    assert np.array_equal(m2, mo.window_mean(mo.time_mean(f))[0])
    both = Mask(ds)([m1, m2])
    ref = mo.apply(mo.apply(f, mo.angle(f)), m2)
    for i, k in enumerate(("v_x", "v_y", "corr", "s2n")):
        assert np.array_equal(both[k], ref[i], equal_nan=True) and np.array_equal(ds[k], f[i], equal_nan=True)
    Mask(ds)([m1, m2], inplace=True)
    assert np.array_equal(ds["s2n"], ref[3], equal_nan=True)
    # the Ngwerere mask recipe (examples/ngwerere/ngwerere.yml:20-37), each in place, in order
    ds, g = mk(), f.copy()
    mref = Mask(ds)
    for name, kw in (("corr", {}), ("minmax", {}), ("rolling", {}), ("outliers", {}), ("variance", {}), ("count", {})):
        getattr(mref, name)(inplace=True, **kw)
        g = mo.apply(g, getattr(mo, name)(g, **kw))
    mref.window_mean(inplace=True, wdw=2, tolerance=0.5, reduce_time=True)
    g = mo.apply(g, mo.window_mean(mo.time_mean(g), wdw=2, tolerance=0.5)[0])
    for i, k in enumerate(("v_x", "v_y", "corr", "s2n")):
        assert np.array_equal(ds[k], g[i], equal_nan=True)
    rep = Mask(ds).window_replace(wdw=1, iter=2)
    assert np.array_equal(np.stack([rep[k] for k in ("v_x", "v_y", "corr", "s2n")]), mo.window_replace(g, wdw=1, iter=2), equal_nan=True)
    single = PivResult({k: f[i, 0].copy() for i, k in enumerate(("v_x", "v_y", "corr", "s2n"))}, {})
    assert np.array_equal(Mask(single).window_nan(), mo.window_nan(f[:, :1])[0])


@pytest.mark.gpu
def test_gpu_device_chain_scale_mask_apply_pack(gpu):
    """Result block stays in HBM: scale to m/s -> masks -> apply -> int16 pack, only the packed block comes back."""
    import ctypes as C

    from oracle import piv_oracle as po
    from pyorc_amd import _lib

    lib = _lib.load()
    T, R, Cc = 30, 12, 14
    f = fields(T, R, Cc, seed=8)
    f[0] *= 4.0
    f[1] *= 4.0                                                                # px / frame
    dt = np.linspace(0.03, 0.05, T)
    n = T * R * Cc
    d_f, d_m, d_p = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for d, b in ((d_f, 16 * n), (d_m, n), (d_p, 8 * n)):
        _lib.check(lib.lspiv_dev_malloc(C.byref(d), b))
    _lib.check(lib.lspiv_memcpy_h2d(d_f, _lib.ptr(f), f.nbytes))
    _lib.check(lib.lspiv_scale_velocity_dev(d_f, T, R * Cc, 0.01, -0.01, _lib.ptr(dt), None))
    g = f.copy()
    g[0], g[1] = mo.scale_velocity(f[0], f[1], 0.01, -0.01, dt)
    for kind, params, ref in ((3, [0.1], lambda x: mo.corr(x)), (0, [0.1, 5.0], lambda x: mo.minmax(x)),
                              (5, [1.0, 0], lambda x: mo.outliers(x)), (2, [0.33], lambda x: mo.count(x))):
        p = np.asarray(params, np.float64)
        _lib.check(lib.lspiv_mask_dev(d_f, T, R, Cc, kind, _lib.ptr(p), len(p), d_m, None))
        _lib.check(lib.lspiv_mask_apply_dev(d_f, T, R, Cc, d_m, int(kind != 2), None))
        g = mo.apply(g, ref(g))
    _lib.check(lib.lspiv_pack_int16_dev(d_f, 4 * n, 0.01, -9999, d_p, None))
    pk = np.empty((4, T, R, Cc), np.int16)
    _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(pk), d_p, pk.nbytes))
    for d in (d_f, d_m, d_p):
        lib.lspiv_dev_free(d)
    assert np.array_equal(pk, po.encode_int16(g)) and (pk != -9999).any()


@pytest.mark.gpu
def test_gpu_rows_golden_filters_and_masks(gpu):
    """HIP results against the committed fixture tests/golden/rows_golden.npz (inputs + oracle outputs of N2 / N3)."""
    import os

    from pyorc_amd import filters
    from pyorc_amd import mask as pm

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rows_golden.npz"))
    fr, f = g["frames"], g["fields"]
    eq = lambda a, b: np.array_equal(a, b, equal_nan=True)
    assert eq(filters.normalize(fr, 2), g["normalize_2"]) and eq(filters.time_diff(fr, thres=3.0, abs=True), g["time_diff"])
    assert eq(filters.minmax(filters.time_diff(fr), -20.0, 35.0), g["minmax"])
    for got, key in ((filters.smooth(fr, 1), "smooth_1"), (filters.smooth(fr, 4), "smooth_4"),
                     (filters.edge_detect(fr, 1, 2), "edge_1_2"), (filters.edge_detect(fr.astype(np.float32) * 0.5 - 30.0, 2, 6), "edge_2_6")):
        assert np.abs(got - g[key]).max() <= 4e-6 * 255
    for name, params in (("minmax", [0.1, 5.0]), ("count", [0.33]), ("corr", [0.3]), ("s2n", [10]), ("outliers", [0.8, 1]),
                         ("variance", [5, 1]), ("rolling", [4, 0.6]), ("window_nan", [0.7, -1, 1, -1, 1]),
                         ("window_mean", [0.5, 1, -2, 2, -2, 2])):
        assert eq(pm.run_mask(f, name, params), g["mask_" + name]), name
    assert eq(pm.time_mean(f), g["time_mean"])


# ---- pinned against the reference's own files ------------------------------------------------------------------------
# examples/ngwerere/ngwerere_piv.nc -> examples/ngwerere/ngwerere_masked.nc, the input and the output of the reference's
# masking notebook (tests/golden/make_golden.py::make_ngwerere_masks holds the recipe and how the fixture was made).

def ngwerere_fixture():
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ngwerere_masks.npz"))
    raw = np.stack([z[k] for k in ("v_x", "v_y", "corr", "s2n")])                       # int16 as stored on disk
    # CF decoding as xarray does it for int16 + scale_factor without add_offset: float32 data, scaled in float32
    f = raw.astype(np.float32) * np.float32(z["scale_factor"])
    f[raw == z["fill"]] = np.nan
    keep = np.unpackbits(z["keep_bits"])[: raw[0].size].reshape(raw[0].shape).astype(bool)
    return raw, f, keep


NOTEBOOK_RECIPE = (("corr", {}), ("minmax", {}), ("rolling", {}), ("outliers", {}), ("variance", {}), ("count", {}))


def test_mask_oracle_reproduces_the_reference_masked_file():
    """PINS the mask oracle: the notebook's chain (corr, minmax, rolling, outliers, variance, count with their defaults,
    then window_mean(wdw=2, tolerance=0.5, reduce_time=True)) applied to ngwerere_piv.nc keeps exactly the 93 824 of
    486 750 vectors that the reference's ngwerere_masked.nc keeps -- no mismatch."""
    raw, f, keep = ngwerere_fixture()
    assert f.shape == (4, 125, 59, 66) and keep.sum() == 93824
    kept_after = []
    for name, kw in NOTEBOOK_RECIPE:
        f = mo.apply(f, getattr(mo, name)(f, **kw))
        kept_after.append(int((~np.isnan(f[0])).sum()))
    f = mo.apply(f, mo.window_mean(mo.time_mean(f), wdw=2, tolerance=0.5)[0])
    assert kept_after == [470651, 272982, 194997, 167523, 167523, 97677]                # every mask bites (variance: no-op)
    for k in range(4):
        assert np.array_equal(~np.isnan(f[k]), keep)
    # N4: re-encoding what is left gives the bytes of ngwerere_masked.nc (kept values are the input's, the rest _FillValue)
    from oracle import piv_oracle as po

    assert np.array_equal(po.encode_int16(f), np.where(keep, raw, np.int16(-9999)))


@pytest.mark.gpu
def test_gpu_masks_reproduce_the_reference_masked_file(gpu):
    """The HIP masks through the reference-shaped wrapper (`Mask`), then the int16 packing kernel: ngwerere_piv.nc in,
    the bytes of ngwerere_masked.nc out."""
    from pyorc_amd.mask import Mask
    from pyorc_amd.velocimetry import PivResult

    raw, f, keep = ngwerere_fixture()
    ds = PivResult({k: f[i].copy() for i, k in enumerate(("v_x", "v_y", "corr", "s2n"))}, {})
    m = Mask(ds)
    for name, kw in NOTEBOOK_RECIPE:
        getattr(m, name)(inplace=True, **kw)
    m.angle(angle_tolerance=0.5 * np.pi)                                                # not in place in the notebook: no effect
    m.window_mean(wdw=2, inplace=True, tolerance=0.5, reduce_time=True)
    out = np.stack([np.asarray(ds[k]) for k in ("v_x", "v_y", "corr", "s2n")])
    assert np.array_equal(~np.isnan(out[0]), keep)
    from pyorc_amd import _lib

    out = np.ascontiguousarray(out, dtype=np.float32)
    packed = np.empty(out.shape, np.int16)
    _lib.check(gpu.lspiv_pack_int16(_lib.ptr(out), out.size, 0.01, -9999, _lib.ptr(packed)))
    assert np.array_equal(packed, np.where(keep, raw, np.int16(-9999)))


REF_NC = "/root/reference/examples/ngwerere/"


@pytest.mark.skipif(not __import__("os").path.exists(REF_NC + "ngwerere_piv.nc"), reason="reference checkout not present (GPU box)")
def test_ngwerere_fixture_is_what_the_reference_files_hold():
    """Provenance of tests/golden/ngwerere_masks.npz (build container only, where /root/reference is mounted): the
    minimal HDF5 reader returns the same int16 variables from the reference's netCDF files as the fixture stores, and
    ngwerere_masked.nc is ngwerere_piv.nc with _FillValue wherever the fixture's keep-mask is False."""
    from oracle import h5min

    raw, _, keep = ngwerere_fixture()
    names = ("v_x", "v_y", "corr", "s2n")
    piv = h5min.read(REF_NC + "ngwerere_piv.nc", names)
    masked = h5min.read(REF_NC + "ngwerere_masked.nc", names)
    for i, k in enumerate(names):
        assert np.array_equal(piv[k], raw[i])
        assert np.array_equal(masked[k], np.where(keep, raw[i], np.int16(-9999)))
