"""GPU tests at the BASELINE.json sizes: 1080p x 1000 pairs (config 2), 64x64@75% (config 3), 4K (config 4).

The oracle cannot finish these sizes in seconds, so they are checked through size-independent properties:
  * a sample of pairs spread over the stack equals the oracle (same gate as the small tests);
  * determinism: two launches give bit-identical results;
  * chunk invariance: the stack processed in time chunks (1-frame halo, cut on the anchors of the grid: 75 pairs at these sizes) equals the
    single launch bit for bit, with the default (time-walking) kernels;
  * round trip: a stack made of circular shifts of one frame returns that shift in every window.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import c_oracle
from pyorc_amd import _lib, window

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel_err(got, ref, floor=0.05):
    with np.errstate(all="ignore"):
        e = np.abs(np.asarray(got, dtype=np.float64) - ref) / np.maximum(np.abs(ref), floor)
    return float(np.nanmax(e)) if np.isfinite(e).any() else 0.0


class DeviceStack:
    def __init__(self, lib, T, H, W, seed):
        self.lib, self.T, self.H, self.W = lib, T, H, W
        self.d = C.c_void_p()
        _lib.check(lib.lspiv_dev_malloc(C.byref(self.d), T * H * W))
        _lib.check(lib.lspiv_synth_particles_dev(self.d, T, H, W, seed, 0.02))

    def run(self, ws, ov, first=0, n_frames=None):
        """PIV of frames [first, first + n_frames): a time chunk whose first pair has index `first` in the stack."""
        n_frames = self.T - first if n_frames is None else n_frames
        nr, nc = window.get_array_shape((self.H, self.W), ws, ov)
        P = n_frames - 1
        d_out = C.c_void_p()
        _lib.check(self.lib.lspiv_dev_malloc(C.byref(d_out), 4 * P * nr * nc * 4))
        try:
            src = C.c_void_p(self.d.value + first * self.H * self.W)
            _lib.check(self.lib.lspiv_piv_pairs_dev_at(src, 0, n_frames, self.H, self.W, ws[0], ws[1], ov[0], ov[1], -1.0,
                                                       first, d_out, None, None))
            out = np.empty((4, P, nr, nc), np.float32)
            _lib.check(self.lib.lspiv_memcpy_d2h(_lib.ptr(out), d_out, out.nbytes))
        finally:
            self.lib.lspiv_dev_free(d_out)
        return out

    def frames(self, first, n):
        a = np.empty((n, self.H, self.W), np.uint8)
        src = C.c_void_p(self.d.value + first * self.H * self.W)
        _lib.check(self.lib.lspiv_memcpy_d2h(_lib.ptr(a), src, a.nbytes))
        return a

    def free(self):
        self.lib.lspiv_dev_free(self.d)


def check_sample(stack, out, ws, ov, starts):
    for s in starts:
        fr = stack.frames(s, 3)
        uo, vo, cmo, sno, cond = c_oracle.piv_pairs(fr, ws, ov, return_cond=True)
        ok = ~c_oracle.exact_tie(cond, cmo)   # every window but exact float64 ties (rescue pass, round 3)
        g = out[:, s:s + 2]
        for k, r in enumerate((uo, vo, cmo, sno)):   # u, v: every window but the exact ties; corr, s2n: ALL windows (the rescue pass never touches them)
            sel = ok if k < 2 else np.ones_like(ok)
            assert np.array_equal(np.isnan(g[k])[sel], np.isnan(r)[sel]), (s, k)
        assert rel_err(g[2], cmo.astype(np.float64)) <= TOL and rel_err(g[3], sno.astype(np.float64)) <= TOL
        assert ok.mean() > 0.9
        assert rel_err(g[0][ok], uo[ok].astype(np.float64)) <= TOL
        assert rel_err(g[1][ok], vo[ok].astype(np.float64)) <= TOL


def test_config2_1080p_1000_pairs(gpu):
    st = DeviceStack(gpu, 1001, 1080, 1920, seed=20260929)
    try:
        ws, ov = (32, 32), (16, 16)
        out = st.run(ws, ov)
        assert out.shape == (4, 1000, 66, 119)
        again = st.run(ws, ov)
        assert np.array_equal(out, again, equal_nan=True)                       # deterministic
        check_sample(st, out, ws, ov, starts=[0, 499, 998])                     # oracle on a spread sample
        from pyorc_amd import window
        assert window.chunk_alignment(ws, (1080, 1920), ov) == 75               # 7 854 windows: the long anchors (round 5)
        for first, n in ((0, 301), (300, 376), (675, 326)):                     # time chunks with a 1-frame halo, cut on
            part = st.run(ws, ov, first=first, n_frames=n)                      # anchors (lspiv_chunk_alignment_grid = 75 pairs)
            assert np.array_equal(part, out[:, first:first + n - 1], equal_nan=True)   # bit-identical, like the reference
        u = out[0]
        assert 2.5 < np.nanmedian(u) < 3.5 and np.isnan(u).mean() < 0.05       # flow 3 + 2 sin(.) px/frame
    finally:
        st.free()


def test_config3_64x64_overlap48(gpu):
    st = DeviceStack(gpu, 1001, 1080, 1920, seed=20260930)                      # BASELINE.json configs[2] at full length
    try:
        ws, ov = (64, 64), (48, 48)
        out = st.run(ws, ov)
        assert out.shape == (4, 1000, 64, 117)
        check_sample(st, out, ws, ov, starts=[0, 498, 998])
        assert np.array_equal(st.run(ws, ov, first=525, n_frames=151), out[:, 525:675], equal_nan=True)   # anchors every 75 pairs on this grid
        u = out[0]
        assert 2.5 < np.nanmedian(u) < 3.5 and np.isnan(u).mean() < 0.05
    finally:
        st.free()


def test_config4_4k(gpu):
    st = DeviceStack(gpu, 1001, 2160, 3840, seed=20260931)                      # BASELINE.json configs[3] at full length
    try:
        ws, ov = (32, 32), (16, 16)
        out = st.run(ws, ov)
        assert out.shape == (4, 1000, 134, 239)
        check_sample(st, out, ws, ov, starts=[0, 998])
        assert np.array_equal(st.run(ws, ov, first=750, n_frames=151), out[:, 750:900], equal_nan=True)
        u = out[0]
        assert 2.5 < np.nanmedian(u) < 3.5 and np.isnan(u).mean() < 0.05
    finally:
        st.free()


def test_1080p_window24_prime_factor_kernels(gpu):
    """The Ngwerere recipe's window (24 x 24 @ overlap 12, the 3 x 8 prime-factor FFT kernels) at the benchmark's frame
    size: 14 151 windows per pair, 300 pairs; oracle on a spread sample, determinism, time chunks, flow statistics."""
    st = DeviceStack(gpu, 301, 1080, 1920, seed=20260932)
    try:
        ws, ov = (24, 24), (12, 12)
        out = st.run(ws, ov)
        assert out.shape == (4, 300, 89, 159)
        assert np.array_equal(out, st.run(ws, ov), equal_nan=True)               # deterministic
        check_sample(st, out, ws, ov, starts=[0, 150, 298])
        assert np.array_equal(st.run(ws, ov, first=150, n_frames=151), out[:, 150:300], equal_nan=True)   # aligned chunk (14 151 windows: anchors every 75 pairs)
        u = out[0]
        assert 2.5 < np.nanmedian(u) < 3.5 and np.isnan(u).mean() < 0.08
    finally:
        st.free()


def test_round_trip_circular_shift_full_frame(gpu):
    """frames[t] = roll(frames[0], t * (dy, dx)): every window must report (dx, dy) to within the bias of a
    non-periodic 32x32 window (content enters / leaves at its edges), on all 7854 windows of all pairs."""
    import pyorc_amd

    rng = np.random.default_rng(5)
    base = (rng.random((1080, 1920)) ** 6 * 255).astype(np.uint8)
    dy, dx = 2, -5
    fr = np.stack([np.roll(base, (t * dy, t * dx), (0, 1)) for t in range(4)])
    u, v, cm, sn = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16))
    # the window content moves rigidly; integer shift => symmetric peak => sub-pixel offset ~ 0
    assert np.nanmax(np.abs(u - dx)) < 0.25 and np.nanmax(np.abs(v - dy)) < 0.25
    assert abs(np.mean(u) - dx) < 0.01 and abs(np.mean(v) - dy) < 0.01
    assert not np.isnan(u).any()


def test_config1_ngwerere_geometry(gpu):
    """BASELINE.json configs[0] geometry (785 x 875, 21 frames, 32x32 @ 50 % -> 48 x 53) through get_piv on the GPU,
    chunked like the reference would on a small machine, against the C oracle."""
    from pyorc_amd import frames as F
    from pyorc_amd.synth import particle_stack

    fr = particle_stack(21, 785, 875, seed=20260927 + 1)
    t = np.arange(21) / 30.0
    ds = F.get_piv(fr, 32, time=t, resolution=0.01, chunksize=5)
    assert ds["v_x"].shape == (20, 48, 53) and np.array_equal(ds.coords["time"], t[1:])
    assert np.array_equal(ds.coords["x"], np.arange(875)[16::16][:53])
    whole = F.get_piv(fr, 32, time=t, resolution=0.01)
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(ds[k], whole[k], equal_nan=True)                    # any chunksize, same bits
    uo, vo, cmo, sno, cond = c_oracle.piv_pairs(fr, (32, 32), (16, 16), return_cond=True)
    ok = ~c_oracle.exact_tie(cond, cmo)
    assert ok.mean() > 0.9
    assert rel_err(ds["corr"], cmo.astype(np.float64)) <= TOL and rel_err(ds["s2n"], sno.astype(np.float64)) <= TOL
    vx_ref = (uo.astype(np.float64) * 0.01 * 30.0)
    assert rel_err(ds["v_x"][ok], vx_ref[ok], floor=0.05 * 0.3) <= TOL


@pytest.mark.parametrize("n,ov,dtype", [(32, 16, np.uint8), (32, 16, np.float32), (64, 48, np.uint8)])
def test_ensemble_on_the_full_width_1080p_grid(gpu, n, ov, dtype):
    """Ensemble mode on window grids as wide as BASELINE.json configs[1] / [2] (119 and 117 columns -- wider than the walking kernels'
    job strips, which no small-frame test is; half the height, to keep the numpy oracle at seconds), a few pairs, every window
    against the oracle of pyorc/velocimetry/ffpiv.py:182-376."""
    from oracle import piv_oracle as po
    from pyorc_amd import frames as F
    from pyorc_amd.synth import particle_stack

    T = 4
    fr = particle_stack(T, 540, 1920, seed=20260940 + n, density=0.03)
    if dtype != np.uint8:
        fr = fr.astype(dtype) * 0.25 + 3.0
    kw = dict(corr_min=0.1, s2n_min=1.5)
    got = F.get_piv(fr, n, overlap=(ov, ov), ensemble_corr=True, **kw)
    ref = po.get_ffpiv(fr, np.ones(T - 1), (n, n), (ov, ov), 1.0, 1.0, ensemble_corr=True, **kw)
    assert got["v_x"].shape == ((1, 32, 119) if n == 32 else (1, 30, 117))
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), k
        assert rel_err(got[k], np.asarray(ref[k], dtype=np.float64)) <= TOL, k


def test_ngwerere_recipe_window_25(gpu):
    """The window the Ngwerere recipe actually asks for (examples/ngwerere/ngwerere.yml: window_size 25 -> 24 after
    round_to_even, overlap int(round(25) / 2) = 12, quirk Q6): 785 x 875 frames, 64 x 71 windows, the 24-point
    FFT kernels, per-timestep and ensemble mode, against the C oracle."""
    from pyorc_amd import frames as F
    from pyorc_amd.synth import particle_stack

    fr = particle_stack(7, 785, 875, seed=20260927 + 7)
    t = np.arange(7) / 30.0
    assert F.resolve_window(25, None) == ((24, 24), (24, 24), (12, 12))
    assert _lib.load().lspiv_kernel_kind(24, 24) == 8
    ds = F.get_piv(fr, 25, time=t, resolution=0.01)
    assert ds["v_x"].shape == (6, 64, 71)
    uo, vo, cmo, sno, cond = c_oracle.piv_pairs(fr, (24, 24), (12, 12), return_cond=True)
    ok = ~c_oracle.exact_tie(cond, cmo)
    assert ok.mean() > 0.8
    assert np.array_equal(np.isnan(ds["corr"]), np.isnan(cmo))
    assert rel_err(ds["corr"], cmo.astype(np.float64)) <= TOL and rel_err(ds["s2n"], sno.astype(np.float64)) <= TOL
    assert rel_err(ds["v_x"][ok], (uo.astype(np.float64) * 0.3)[ok], floor=0.05 * 0.3) <= TOL
    assert rel_err(ds["v_y"][ok], (vo.astype(np.float64) * 0.3)[ok], floor=0.05 * 0.3) <= TOL
    ens = F.get_piv(fr, 25, time=t, resolution=0.01, ensemble_corr=True)
    assert ens["v_x"].shape == (1, 64, 71) and np.isfinite(ens["v_x"]).mean() > 0.9
    assert abs(float(np.nanmedian(ens["v_x"])) - float(np.nanmedian(ds["v_x"]))) < 0.05


def test_device_chain_on_a_stack_beyond_2_31_elements(gpu):
    """The rows around the hot path on an HBM-resident stack of 1110 x 1080 x 1920 uint8 frames = 2.30 G elements
    (9.2 GB as float32): element indices pass 2^31 and 2^32 bytes.  numpy cannot hold the oracle of the whole stack, so frames from
    the start, from either side of the 2^31-element mark (frame 1035 / 1036) and from the end are compared with the oracles of
    range (the whole stack), normalize, time_diff, minmax, smooth and the orthoprojection (group means) -- bit-exact, blur within 4e-6 of the range, as in the
    small tests -- and get_piv runs on the projected float32 stack."""
    from oracle import filters_oracle as fo, project_oracle as pro
    from pyorc_amd import DeviceFrames, filters
    from pyorc_amd import frames as F
    from pyorc_amd.project import Projection
    from pyorc_amd.synth import projection_maps

    T, H, W = 1110, 1080, 1920
    cam = DeviceFrames.empty((T, H, W), np.uint8)
    _lib.check(gpu.lspiv_synth_particles_dev(cam.c_ptr, T, H, W, 20260950, 0.02))
    _lib.check(gpu.lspiv_synchronize())
    picks = [0, 1035, 1036, T - 2]
    full = cam.to_host()                                         # 2.3 GB of bytes: numpy can still reduce those through time
    host = {k: full[k:k + 2] for k in picks}
    assert np.array_equal(filters.range(cam), fo.time_range(full))
    # the host entry point on the same 2.3 GB (pageable memory -> pinned staging slots -> HBM in sub-batches cut on the anchors)
    # against the HBM-resident stack: the same bits
    import pyorc_amd
    for a, b in zip(pyorc_amd.piv_pairs(full, (32, 32), (16, 16)), pyorc_amd.piv_pairs(cam, (32, 32), (16, 16))):
        assert a.shape == (T - 1, 66, 119) and np.array_equal(a, b, equal_nan=True)
    # ... and a float64 host stack of 4.6 GB (what project_numpy hands over; narrowed to float32 by the staging threads, past
    # 2^32 bytes) against the same frames as float32 in HBM
    f64 = full[:280].astype(np.float64)
    dev32 = DeviceFrames.from_host(full[:280].astype(np.float32))
    for a, b in zip(pyorc_amd.piv_pairs(f64, (32, 32), (16, 16)), pyorc_amd.piv_pairs(dev32, (32, 32), (16, 16))):
        assert np.array_equal(a, b, equal_nan=True)
    del f64, dev32
    sub = full[:230].copy()                                      # for the fixed chain below

    norm = filters.normalize(cam, 15)
    mean = full[::round(T / 15)].mean(axis=0).astype("float32")
    del full
    for k in picks:
        assert np.array_equal(norm[k:k + 2].to_host(), fo.normalize_with_mean(host[k], mean)), k

    diff = filters.time_diff(cam, thres=2.0, abs=True)
    assert diff.shape == (T - 1, H, W)
    for k in picks:
        assert np.array_equal(diff[k:k + 1].to_host(), fo.time_diff(host[k], thres=2.0, abs=True)), k
    clipped = filters.minmax(diff, min=1.0, max=40.0)
    for k in picks:
        assert np.array_equal(clipped[k:k + 1].to_host(), fo.minmax(fo.time_diff(host[k], thres=2.0, abs=True), 1.0, 40.0)), k
    del diff, clipped

    sm = filters.smooth(cam, 2)
    for k in picks:
        ref = fo.smooth(host[k][:1], 2)
        assert np.abs(sm[k:k + 1].to_host() - ref).max() <= 4e-6 * 255, k
    del sm

    dst = (810, 1440)                                             # 3/4 of the camera's resolution: 22 % of the cells are group means
    idx_img, mask, src_idx, uidx, norm_idx = projection_maps((H, W), dst, tilt=0.1, seed=1)
    p = Projection((H, W), dst, idx_img, mask, src_idx=src_idx, uidx=uidx, norm_idx=norm_idx)
    assert not p.nearest_only                                     # cells with group means: float32 ortho frames
    ortho = p.project_frames(norm, keep_uint8=False)
    assert ortho.dtype == np.float32 and ortho.shape == (T,) + dst and len(uidx) > 200000
    for k in picks:
        ref = pro.project_frames(norm[k:k + 1].to_host(), dst, idx_img, mask, src_idx=src_idx, uidx=uidx, norm_idx=norm_idx)
        assert np.array_equal(ortho[k:k + 1].to_host().astype(np.float64), ref), k
    # the fixed chain (pipeline.CameraToVelocity) on 230 frames of the same camera stack: one piece, streamed in time chunks, and
    # the three stages one by one -- the same bits
    from pyorc_amd.pipeline import CameraToVelocity

    with CameraToVelocity((H, W), dst, idx_img, mask, src_idx, uidx, norm_idx, window_size=(32, 32), overlap=(16, 16), normalize_samples=15) as chain:
        one = chain.run(sub, streamed=False)
        streamed = chain.run(sub, streamed=True)
    staged = pyorc_amd.piv_pairs(p.project_frames(filters.normalize(DeviceFrames.from_host(sub), 15), keep_uint8=False), (32, 32), (16, 16))
    for a, b, c in zip(one, streamed, staged):
        assert a.shape == (229, 49, 89) and np.array_equal(a, b, equal_nan=True) and np.array_equal(a, c, equal_nan=True)
    p.close()
    del norm, sub

    ds = F.get_piv(ortho, 32, time=np.arange(T) / 30.0, resolution=0.01)
    assert ds["v_x"].shape == (T - 1, 49, 89)
    for k in (0, 1035, T - 2):
        uo, vo, cmo, sno, cond = c_oracle.piv_pairs(ortho[k:k + 2].to_host(), (32, 32), (16, 16), return_cond=True)
        ok = ~c_oracle.exact_tie(cond, cmo)
        assert np.array_equal(np.isnan(ds["corr"][k]), np.isnan(cmo[0])) and rel_err(ds["corr"][k], cmo[0].astype(np.float64)) <= TOL
        assert rel_err(ds["v_x"][k][ok[0]], (uo[0].astype(np.float64) * 0.3)[ok[0]], floor=0.05 * 0.3) <= TOL, k
        assert rel_err(ds["v_y"][k][ok[0]], (vo[0].astype(np.float64) * 0.3)[ok[0]], floor=0.05 * 0.3) <= TOL, k


def test_rows_at_1080p_new_kernels_equal_the_ones_they_replace(gpu, monkeypatch):
    """The two row kernels that came last in round 6, at the size the bench line quotes them on (1080 x 1920 camera frames, the 810 x 1440
    ortho grid of the bench's lens + perspective), through their size-independent property: the SAME bits as the kernels they replace --
    `blur_strip4_kernel` (four columns per lane, integer row pass; 7.5 strips of 256 columns per row, 33.75 strips of 32 rows) against the
    one-column float kernel for every unrolled radius, `remap_fused_kernel` (undistortion + warp in one kernel, 23 x 51 tiles) against the
    two remap passes -- and a sample of frames against the oracles."""
    from oracle import filters_oracle as fo
    from oracle import project_oracle as pj
    from pyorc_amd import DeviceFrames, filters
    from pyorc_amd.project import ProjectionCV
    from pyorc_amd.synth import particle_stack

    fr = particle_stack(9, 1080, 1920, seed=77)
    fr[3, 500:600, 900:1100] = 255                                     # a saturated patch: the largest sums of the integer row pass
    dev = DeviceFrames.from_host(fr)
    for name, args in (("smooth", (1,)), ("smooth", (2,)), ("smooth", (3,)), ("edge_detect", (1, 2)), ("edge_detect", (1, 3)), ("edge_detect", (2, 3))):
        monkeypatch.delenv("LSPIV_BLUR_ONE_COLUMN", raising=False)
        four = getattr(filters, name)(dev, *args).to_host()
        monkeypatch.setenv("LSPIV_BLUR_ONE_COLUMN", "1")
        one = getattr(filters, name)(dev, *args).to_host()
        assert np.array_equal(four.view(np.uint32), one.view(np.uint32)), (name, args)
        ref = getattr(fo, name)(fr[3:4], *args)
        assert np.abs(four[3:4] - ref).max() <= 4e-6 * 255, (name, args)
    monkeypatch.delenv("LSPIV_BLUR_ONE_COLUMN", raising=False)

    K = np.array([[1500.0, 0, 960.0], [0, 1500.0, 540.0], [0, 0, 1]])
    dist = np.array([-0.12, 0.03, 0.001, -0.0005, 0.0])
    M = np.array([[0.78, 0.05, -20.0], [0.01, 0.80, -15.0], [1.5e-5, 4.0e-5, 1.0]])
    monkeypatch.setenv("LSPIV_PROJECT_CV_TWO_PASS", "1")
    two = ProjectionCV((1080, 1920), (810, 1440), K, dist, M)
    monkeypatch.delenv("LSPIV_PROJECT_CV_TWO_PASS")
    one = ProjectionCV((1080, 1920), (810, 1440), K, dist, M)
    try:
        a, b = one.project_frames(dev).to_host(), two.project_frames(dev).to_host()
        assert a.dtype == np.uint8 and np.array_equal(a, b)
        assert np.array_equal(a[3], pj.project_cv(fr[3:4], K, dist, M, (810, 1440))[0])
    finally:
        one.close(); two.close()
