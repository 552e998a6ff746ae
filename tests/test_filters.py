"""N2 element-wise filters: oracle behaviour on CPU, bit-exact parity on the GPU."""
import os

import numpy as np
import pytest

from oracle import filters_oracle as fo
from pyorc_amd.synth import particle_stack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_filters_follow_reference_arithmetic():
    fr = particle_stack(31, 48, 64, seed=2)
    n = fo.normalize(fr, samples=15)
    assert n.dtype == np.uint8 and n.shape == fr.shape
    assert (n.reshape(31, -1).max(axis=1) == 255).all() and (n.reshape(31, -1).min(axis=1) == 0).all()
    iv = round(31 / 15)
    mean = fr[::iv].mean(axis=0).astype("float32")
    red = fr.astype("float32") - mean
    k = 7
    exp = ((red[k] - red[k].min()) / (red[k].max() - red[k].min()) * 255).astype("uint8")
    assert np.array_equal(n[k], exp)
    d = fo.time_diff(fr, thres=3.0)
    assert d.dtype == np.float32 and d.shape == (30, 48, 64) and ((d == 0) | (d > 3.0)).all()
    raw = np.diff(fr.astype(np.float32), axis=0)
    assert np.array_equal(d, np.where(raw > 3.0, raw, 0))
    f = raw.copy(); f[0, 0, 0] = np.nan
    m = fo.minmax(f, -5, 5)
    assert np.isnan(m[0, 0, 0]) and np.nanmin(m) == -5 and np.nanmax(m) == 5
    with pytest.raises(AssertionError):
        fo.normalize(fr[:5], samples=15)
    rr = fo.reduce_rolling(np.array([[[10, 0]], [[20, 0]], [[60, 0]], [[10, 0]]], np.uint8), samples=2)
    # rolling means 15, 40, 35: excesses 5, 20, 0 -> each frame stretched to its own maximum; mean 0 -> 0; frame 0 has no window
    assert rr.dtype == np.uint8 and rr[:, 0, :].tolist() == [[0, 0], [255, 0], [255, 0], [0, 0]]
    r = fo.time_range(np.array([[[3, 200]], [[9, 7]], [[4, 255]]], np.uint8))
    assert r.dtype == np.uint8 and r.tolist() == [[6, 248]]                       # max - min through time, input dtype kept


def test_filter_wrappers_validate_before_touching_the_device():
    """Argument errors of the reference's filters come from the host wrapper, with the reference's messages, whether or
    not a GPU is present (pyorc/api/frames.py:296-298, 397-398)."""
    from pyorc_amd import filters

    fr = particle_stack(6, 16, 24, seed=1)
    with pytest.raises(AssertionError, match="smaller than requested rolling of 25 samples"):
        filters.reduce_rolling(fr, samples=25)
    with pytest.raises(ValueError, match="uint8"):
        filters.reduce_rolling(fr.astype(np.float32), samples=2)
    with pytest.raises(AssertionError, match="too small to provide 15 samples"):
        filters.normalize(fr[:5], samples=15)
    with pytest.raises(ValueError):
        filters.range(fr[0])                                                   # not a (T, H, W) stack


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
def test_gpu_time_diff_bit_exact(gpu, dtype):
    from pyorc_amd import filters

    fr = particle_stack(9, 130, 171, seed=4)
    fr = fr if dtype == np.uint8 else fr.astype(dtype) * 0.37 - 3.0
    if dtype != np.uint8:
        fr[3, 5, 7] = np.nan
    for thres, ab in ((0.0, False), (2.5, False), (-1.0, True)):
        got = filters.time_diff(fr, thres, ab)
        assert got.dtype == np.float32 and np.array_equal(got, fo.time_diff(fr, thres, ab))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,shape", [(np.uint8, (9, 130, 176)), (np.uint8, (5, 33, 41)), (np.float32, (9, 130, 171)),
                                         (np.float64, (4, 50, 61)), (np.uint8, (1, 16, 16))])
def test_gpu_range_bit_exact(gpu, dtype, shape):
    """Frames.range (pyorc/api/frames.py:364-379): max - min through time in the frames' own dtype; the 16-pixel uint8
    path (frame size a multiple of 16) and the scalar one, NaN skipped for float frames, an all-NaN pixel stays NaN."""
    from pyorc_amd import filters

    fr = particle_stack(*shape, seed=14)
    if dtype != np.uint8:
        fr = fr.astype(dtype) * 0.37 - 3.0
        fr[shape[0] // 2, 5, 7] = np.nan
        fr[:, 6, 8] = np.nan
    got = filters.range(fr)
    ref = fo.time_range(fr)
    assert got.dtype == fr.dtype and got.shape == fr.shape[1:] and np.array_equal(got, ref, equal_nan=True)
    if dtype != np.uint8:
        assert np.isnan(got[6, 8]) and not np.isnan(got[5, 7])


@pytest.mark.gpu
@pytest.mark.parametrize("shape,samples", [((40, 130, 176), 25), ((31, 57, 83), 7), ((9, 16, 16), 9), ((12, 33, 41), 1),
                                           ((300, 64, 96), 25)])
def test_gpu_reduce_rolling_bit_exact(gpu, shape, samples):
    """Frames.reduce_rolling (pyorc/api/frames.py:381-407): float64 arithmetic on exact integer window sums; frame sizes
    with and without a 16-pixel tail, several time segments with their halo, a black region (rolling mean 0), a frame
    brighter nowhere than its rolling mean (maximum 0)."""
    from pyorc_amd import filters

    fr = particle_stack(*shape, seed=15)
    fr[:, :5, :7] = 0                                   # rolling mean 0 -> 0
    if shape[0] > samples + 2:
        fr[samples + 1] = 0                             # nothing above the rolling mean: 0 / 0 -> 0
    got = filters.reduce_rolling(fr, samples)
    ref = fo.reduce_rolling(fr, samples)
    assert got.dtype == np.uint8 and np.array_equal(got, ref)
    assert (got[: samples - 1] == 0).all() and got[samples - 1:].max() == (255 if samples > 1 else 0)   # samples = 1: frame - itself
    with pytest.raises(AssertionError):
        filters.reduce_rolling(fr[: samples - 1] if samples > 1 else fr[:0], samples)


@pytest.mark.gpu
def test_gpu_minmax_and_normalize_bit_exact(gpu):
    from pyorc_amd import _lib, filters

    rng = np.random.default_rng(3)
    f = (rng.standard_normal((4, 77, 93)) * 6).astype(np.float32)
    f[1, 2, 3] = np.nan
    for lo, hi in ((-5, 5), (-np.inf, 2.0), (0.0, np.inf)):
        assert np.array_equal(filters.minmax(f, lo, hi), fo.minmax(f, lo, hi), equal_nan=True)
    fr = particle_stack(31, 120, 160, seed=6)
    fr[5] = 17  # a constant frame: 0/0 -> 0
    for samples in (15, 4, 31):
        assert np.array_equal(filters.normalize(fr, samples), fo.normalize(fr, samples)), samples
    with pytest.raises(AssertionError):
        filters.normalize(fr[:5], 15)
    with pytest.raises(ValueError):
        filters.normalize(fr.astype(np.float32))


@pytest.mark.gpu
def test_gpu_recipe_chain_normalize_project_piv(gpu):
    """The Ngwerere recipe order (examples/ngwerere/ngwerere.yml:5-11, minus the cv2 edge filter): normalize ->
    project -> get_piv, every stage on the GPU, against the oracle chain."""
    import pyorc_amd
    from oracle import c_oracle, project_oracle as pro
    from pyorc_amd import filters
    from pyorc_amd.project import Projection
    from pyorc_amd.synth import projection_maps

    src, dst = (240, 320), (128, 160)
    cam = particle_stack(16, src[0], src[1], seed=8, density=0.04)
    cam = (cam * 0.6 + 60).astype(np.uint8)  # static background offset that normalize removes
    maps = projection_maps(src, dst, tilt=0.25, seed=2)
    norm = filters.normalize(cam, samples=15)
    assert np.array_equal(norm, fo.normalize(cam, 15))
    p = Projection(src, dst, *maps)
    ortho = p.project_frames(norm)
    p.close()
    ref_ortho = pro.project_frames(fo.normalize(cam, 15), dst, *maps)
    assert np.array_equal(ortho.astype(np.float64), ref_ortho)
    u, v, cm, sn = pyorc_amd.piv_pairs(ortho, (32, 32), (16, 16))
    uo, vo, cmo, sno, cond = c_oracle.piv_pairs(ref_ortho, (32, 32), (16, 16), return_cond=True)
    ok = ~c_oracle.exact_tie(cond, cmo)   # all windows but exact float64 ties (float64 rescue pass)
    e = lambda g, r: float(np.nanmax(np.abs(g - r) / np.maximum(np.abs(r), 0.05)))
    assert ok.mean() > 0.99 and e(cm, cmo) <= 1e-4 and e(u[ok], uo[ok]) <= 1e-4 and e(v[ok], vo[ok]) <= 1e-4


@pytest.mark.gpu
def test_gpu_camera_to_velocity_chain_equals_stage_by_stage(gpu):
    """pipeline.CameraToVelocity (one H2D, every stage a *_dev call, one D2H) == the stand-alone mirrors, bit for bit."""
    import pyorc_amd
    from oracle import piv_oracle as po
    from pyorc_amd import filters
    from pyorc_amd.pipeline import CameraToVelocity
    from pyorc_amd.project import Projection
    from pyorc_amd.synth import projection_maps

    src, dst = (240, 320), (128, 160)
    cam = (particle_stack(16, src[0], src[1], seed=18, density=0.04) * 0.6 + 50).astype(np.uint8)
    maps = projection_maps(src, dst, tilt=0.25, seed=5)
    p = Projection(src, dst, *maps)
    for samples in (None, 15):
        staged = filters.normalize(cam, samples) if samples else cam
        ref = pyorc_amd.piv_pairs(p.project_frames(staged), (32, 32), (16, 16))
        with CameraToVelocity(src, dst, *maps, window_size=(32, 32), overlap=(16, 16), normalize_samples=samples) as chain:
            got = chain.run(cam)
            for a, b in zip(ref, got):
                assert np.array_equal(a, b, equal_nan=True)
            pk = chain.run(cam, packed=True)
            for a, b in zip(ref, pk):
                assert b.dtype == np.int16 and np.array_equal(b, po.encode_int16(a))
            assert chain.run(cam[:9])[0].shape == (8, 7, 9)   # buffers are reused for a shorter chunk
            with pytest.raises(ValueError):
                chain.run(cam.astype(np.float32))
    # the Ngwerere frames recipe: normalize -> edge_detect(1, 2) -> minmax(-5, 5) -> project -> get_piv
    staged = filters.minmax(filters.edge_detect(filters.normalize(cam, 15), 1, 2), -5, 5)
    ref = pyorc_amd.piv_pairs(p.project_frames(staged), (32, 32), (16, 16))
    with CameraToVelocity(src, dst, *maps, normalize_samples=15, edge_detect=(1, 2), minmax=(-5, 5)) as chain:
        for a, b in zip(ref, chain.run(cam)):
            assert np.array_equal(a, b, equal_nan=True)
    p.close()


@pytest.mark.gpu
def test_gpu_streamed_chain_equals_one_piece(gpu):
    """CameraToVelocity.run(streamed=True): sampled frames first, then time chunks cut on the kernels' anchors, upload of
    chunk k+1 overlapped with the kernels of chunk k -- the bits of the one-piece run, for every recipe, float and packed."""
    from pyorc_amd.pipeline import CameraToVelocity
    from pyorc_amd.synth import projection_maps

    src, dst = (120, 160), (96, 128)
    cam = (particle_stack(83, src[0], src[1], seed=28, density=0.04) * 0.6 + 50).astype(np.uint8)
    maps = projection_maps(src, dst, tilt=0.2, seed=6)
    for kw in (dict(), dict(normalize_samples=15), dict(normalize_samples=7, edge_detect=(1, 2), minmax=(-5, 5)),
               dict(normalize_samples=15, window_size=(20, 20), overlap=(10, 10)), dict(edge_detect=(1, 3))):
        with CameraToVelocity(src, dst, *maps, **kw) as chain:
            assert len(chain._chunk_bounds(82, 8)) >= 4
            for packed in (False, True):
                ref = chain.run(cam, packed=packed, streamed=False)
                for n_chunks in (8, 3):
                    got = chain.run(cam, packed=packed, streamed=True, n_chunks=n_chunks)
                    for a, b in zip(ref, got):
                        assert a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True), (kw, packed, n_chunks)
            short = chain.run(cam[:12], streamed=True)           # fewer pairs than one anchor: falls back to one piece
            for a, b in zip(chain.run(cam[:12], streamed=False), short):
                assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.gpu
def test_gpu_get_piv_on_device_stacks_equals_the_chain_and_the_host_path(gpu):
    """The accessor-shaped call on HBM-resident stacks: DeviceFrames -> filters.normalize -> Projection.project_frames ->
    frames.get_piv(engine="hip") never bounces through the host, and returns, bit for bit, what the fixed chain
    (pipeline.CameraToVelocity) and the host-fed mirrors return -- per-timestep (several chunk sizes) and ensemble."""
    import pyorc_amd
    from pyorc_amd import DeviceFrames, filters, frames as F
    from pyorc_amd.pipeline import CameraToVelocity
    from pyorc_amd.project import Projection
    from pyorc_amd.synth import projection_maps

    src, dst = (240, 320), (128, 160)
    cam = (particle_stack(40, src[0], src[1], seed=28, density=0.04) * 0.6 + 50).astype(np.uint8)
    maps = projection_maps(src, dst, tilt=0.25, seed=6)
    t = np.arange(40) / 25.0
    p = Projection(src, dst, *maps)
    # host-fed: every stage returns a numpy stack
    host = F.get_piv(p.project_frames(filters.normalize(cam, 15)), 32, time=t, resolution=0.02)
    # device-resident: one H2D of the camera frames, results back
    d_cam = DeviceFrames.from_host(cam)
    assert d_cam.shape == cam.shape and len(d_cam[3:10]) == 7 and np.array_equal(d_cam[3:10].to_host(), cam[3:10])
    d_norm = filters.normalize(d_cam, 15)
    d_ortho = p.project_frames(d_norm)
    assert isinstance(d_ortho, DeviceFrames) and d_ortho.dtype == np.float32 and d_ortho.shape == (40,) + dst
    assert np.array_equal(d_norm.to_host(), filters.normalize(cam, 15))
    dev = F.get_piv(d_ortho, 32, time=t, resolution=0.02)
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(dev[k], host[k], equal_nan=True), k
    assert np.array_equal(dev.coords["time"], host.coords["time"])
    for cs in (5, 26, 30):
        part = F.get_piv(d_ortho, 32, time=t, resolution=0.02, chunksize=cs)
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(part[k], host[k], equal_nan=True), (cs, k)
    with CameraToVelocity(src, dst, *maps, window_size=(32, 32), overlap=(16, 16), normalize_samples=15) as chain:
        u, v, cm, sn = chain.run(cam)
    assert np.array_equal(cm, dev["corr"], equal_nan=True) and np.array_equal(sn, dev["s2n"], equal_nan=True)
    dt = np.diff(t)[:, None, None]
    assert np.array_equal((u * 0.02 / dt).astype(np.float32), dev["v_x"], equal_nan=True)
    # the other filters and ensemble mode on device stacks
    e_host = filters.minmax(filters.edge_detect(filters.normalize(cam, 15), 1, 2), -5, 5)
    e_dev = filters.minmax(filters.edge_detect(d_norm, 1, 2), -5, 5)
    assert np.array_equal(e_dev.to_host(), e_host)
    assert np.array_equal(filters.time_diff(d_cam, 2.0).to_host(), filters.time_diff(cam, 2.0))
    assert np.array_equal(filters.smooth(d_cam, 2).to_host(), filters.smooth(cam, 2))
    assert np.array_equal(filters.range(d_cam), filters.range(cam))
    ens_h = F.get_piv(p.project_frames(e_host), 32, time=t, resolution=0.02, ensemble_corr=True, corr_min=0.1, s2n_min=1.5)
    ens_d = F.get_piv(p.project_frames(e_dev), 32, time=t, resolution=0.02, ensemble_corr=True, corr_min=0.1, s2n_min=1.5)
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(ens_d[k], ens_h[k], equal_nan=True), k
    u2, v2, _, _ = pyorc_amd.piv_pairs(d_ortho, (32, 32), (16, 16))
    assert np.array_equal(u2, u, equal_nan=True)
    p.close()


def test_oracle_gaussian_blur_matches_scipy_mirror_correlation():
    from scipy.ndimage import correlate1d

    img = np.random.default_rng(0).random((37, 53)).astype(np.float32)
    for ks in (1, 3, 5, 7, 9, 15, 31):
        k = fo.gaussian_kernel(ks).astype(np.float64)
        assert abs(k.sum() - 1) < 1e-6 and len(k) == ks and np.allclose(k, k[::-1])
        ref = correlate1d(correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        assert np.abs(fo.gaussian_blur(img, ks) - ref).max() < 5e-7
    assert fo.gaussian_kernel(5).tolist() == [0.0625, 0.25, 0.375, 0.25, 0.0625]      # OpenCV's fixed table
    e = fo.edge_detect(img[None], 1, 2)
    assert e.shape == (1, 37, 53) and e.dtype == np.float32 and abs(float(e.mean())) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
def test_gpu_smooth_and_edge_detect(gpu, dtype):
    from pyorc_amd import filters

    fr = particle_stack(3, 70, 131, seed=21, density=0.05)
    fr = fr if dtype == np.uint8 else fr.astype(dtype) * 0.41 - 7.0
    for wdw in (1, 2, 3, 5, 15):
        got = filters.smooth(fr, wdw)
        ref = fo.smooth(fr, wdw)
        assert got.dtype == np.float32 and np.abs(got - ref).max() <= 2e-6 * max(1.0, float(np.abs(ref).max()))
    for w1, w2 in ((1, 2), (1, 3), (2, 7)):
        got = filters.edge_detect(fr, w1, w2)
        ref = fo.edge_detect(fr, w1, w2)
        assert np.abs(got - ref).max() <= 4e-6 * max(1.0, float(np.abs(fr).max()))
    assert np.abs(filters.smooth(fr[0], 1) - fo.gaussian_blur(fr[0], 3)).max() < 1e-4
    tiny = fr[:, :5, :3]                                                          # smaller than the halo: reflect101 wraps
    assert np.abs(filters.smooth(tiny, 3) - fo.smooth(tiny, 3)).max() <= 2e-6 * max(1.0, float(np.abs(tiny).max()))


@pytest.mark.gpu
def test_gpu_normalize_without_the_division_keeps_the_reference_integers(gpu, monkeypatch):
    """Round 6: the stretch pass multiplies by RN(1 / span) and divides only where the product lies within rounding distance of an
    integer.  Frames of every span 1 .. 255 around a constant mean (the quotient times 255 is then an exact integer for many pixels:
    every one of them must take the division), random means, a constant frame (0 / 0) -- the bytes of the reference's float32
    expression (oracle), and of the all-division pass the library ran before."""
    import subprocess
    import sys as _sys

    from pyorc_amd import filters

    rng = np.random.default_rng(12)
    T, H, W = 45, 48, 256
    const_mean = np.empty((T, H, W), np.uint8)
    for t in range(T):
        span = 1 + (t * 6) % 255
        lo = int(rng.integers(0, 256 - span))
        const_mean[t] = rng.integers(lo, lo + span + 1, (H, W))
        const_mean[t, 0, 0], const_mean[t, 0, 1] = lo, lo + span       # the frame's range is exactly `span`
    const_mean[::3] = 100                                              # the sampled frames (interval 3 for 15 samples): mean plane = 100
    rnd = (rng.random((T, H, W)) * rng.integers(2, 256, (T, 1, 1))).astype(np.uint8)
    rnd[7] = 31                                                        # a constant frame away from the mean
    for fr in (const_mean, rnd):
        got = filters.normalize(fr, 15)
        assert np.array_equal(got, fo.normalize(fr, 15))
    # the same stacks through the all-division pass (the switch is read once per process)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from pyorc_amd import filters; a = np.load(sys.argv[1]);"
            "np.save(sys.argv[2], np.stack([filters.normalize(f, 15) for f in a]))" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "in.npy"), np.stack([const_mean, rnd]))
        subprocess.check_call([_sys.executable, "-c", code, os.path.join(d, "in.npy"), os.path.join(d, "out.npy")],
                              env=dict(os.environ, LSPIV_NORM_DIVIDE="1"))
        old = np.load(os.path.join(d, "out.npy"))
    assert np.array_equal(old[0], filters.normalize(const_mean, 15)) and np.array_equal(old[1], filters.normalize(rnd, 15))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(45, 48, 256), (9, 270, 480), (4, 1080, 1920), (3, 64, 1024), (1, 16, 16), (13, 100, 176)])
def test_gpu_normalize_in_one_pass_is_the_two_passes(gpu, monkeypatch, shape):
    """Round 6, opt-in (LSPIV_NORM_ONE_PASS=1; measured slower, see filters.hip): norm_onepass_kernel reads every frame ONCE -- the blocks
    that own the slices of a frame exchange their minima (device-coherent stores, counters per frame) while they hold the frame in
    registers.  The bytes of the default two-pass path and of the oracle: frame counts around the groups of four and the two-group trips,
    one slice and 253, a constant frame, device-resident and host stacks; and when the exchange gives up (LSPIV_NORM_ONEPASS_FAIL=1 --
    what a timeout on a device shared with another long kernel does) the guarded passes behind it deliver the same bytes."""
    from pyorc_amd import DeviceFrames, filters

    rng = np.random.default_rng(shape[0] * 7 + shape[2])
    T = shape[0]
    fr = (rng.random(shape) * rng.integers(2, 256, (T, 1, 1))).astype(np.uint8)
    if T > 2:
        fr[2] = 77
    samples = min(15, T)
    ref = fo.normalize(fr, samples)
    monkeypatch.delenv("LSPIV_NORM_ONE_PASS", raising=False)
    monkeypatch.delenv("LSPIV_NORM_ONEPASS_FAIL", raising=False)
    two = filters.normalize(fr, samples)
    monkeypatch.setenv("LSPIV_NORM_ONE_PASS", "1")
    one = filters.normalize(fr, samples)
    one_dev = filters.normalize(DeviceFrames.from_host(fr), samples).to_host()
    monkeypatch.setenv("LSPIV_NORM_ONEPASS_FAIL", "1")
    fell_back = filters.normalize(DeviceFrames.from_host(fr), samples).to_host()
    assert np.array_equal(two, ref) and np.array_equal(one, ref) and np.array_equal(one_dev, ref) and np.array_equal(fell_back, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
def test_gpu_edge_detect_with_the_clip_in_its_store_is_edge_detect_then_minmax(gpu, dtype):
    """Round 6: lspiv_edge_detect_clip_dev = Frames.edge_detect followed by Frames.minmax (the recipe's order) in one pass.  The bits of the
    two calls, for every kernel family (unrolled radii, run-time radii), NaN samples (they propagate through both) and one-sided limits;
    without limits it is lspiv_edge_detect_dev."""
    import ctypes as C

    from pyorc_amd import _lib, filters
    from pyorc_amd.device import DeviceFrames

    fr = particle_stack(4, 70, 132, seed=23, density=0.05)
    if dtype != np.uint8:
        fr = fr.astype(dtype) * 0.41 - 7.0
        fr[1, 5, 7] = np.nan
    d = DeviceFrames.from_host(fr) if dtype != np.float64 else None
    T, H, W = fr.shape
    inf = float("inf")
    for w1, w2 in ((1, 2), (1, 3), (2, 3), (2, 7), (4, 5)):
        edge = filters.edge_detect(fr, w1, w2)
        for lo, hi in ((-5.0, 5.0), (-inf, 0.25), (0.0, inf), (-inf, inf)):
            ref = filters.minmax(edge, lo, hi)
            if d is None:
                continue                                              # float64 stacks have no device-resident form: the host mirror below
            out = DeviceFrames.empty((T, H, W), np.float32)
            _lib.check(gpu.lspiv_edge_detect_clip_dev(d.c_ptr, _lib.DTYPE_CODES[np.dtype(d.dtype)], T, H, W, 2 * w1 + 1, 2 * w2 + 1, lo, hi, out.c_ptr, None))
            got = out.to_host()
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (w1, w2, lo, hi)
    with pytest.raises(_lib.LspivError):
        d8 = DeviceFrames.from_host(particle_stack(2, 16, 16, seed=1))
        out = DeviceFrames.empty((2, 16, 16), np.float32)
        _lib.check(gpu.lspiv_edge_detect_clip_dev(d8.c_ptr, 0, 2, 16, 16, 3, 5, float("nan"), 1.0, out.c_ptr, None))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(70, 500), (33, 256), (64, 252), (5, 8), (3, 4), (9, 12), (20, 16), (130, 1028), (40, 260), (31, 264)])
def test_gpu_four_column_integer_blur_equals_the_one_column_float_kernel(gpu, monkeypatch, shape):
    """Round 6: uint8 frames whose width is a multiple of four go through blur_strip4_kernel -- four columns per lane, the row pass on the
    packed bytes (v_alignbyte + v_dot4 with OpenCV's integer taps 1 2 1 / 1 4 6 4 1 / 2 7 14 18 14 7 2), the column pass on integer-valued
    floats two columns per instruction.  Every float operation of the one-column kernel is exact on uint8 input, so both give the exact
    rational: the same bits, for every unrolled radius, on widths with a partial last strip (the lane at an edge mirrors the dword beyond it
    with v_perm_b32 from the bytes it loaded), frames of one strip whose first lane is the left and third lane the right edge (12 columns),
    narrower ones (they stay with the one-column kernel), heights around the 32-row strips and below the halo (BORDER_REFLECT_101 wraps);
    and within the oracle's tolerance."""
    from pyorc_amd import filters

    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    fr = (rng.random((3,) + shape) * 256).astype(np.uint8)
    fr[0, 0, :] = 255
    fr[1, :, -1] = 255
    fr[2] = 255                                                            # the largest sums everywhere
    for wdw in (1, 2, 3):
        monkeypatch.delenv("LSPIV_BLUR_ONE_COLUMN", raising=False)
        got = filters.smooth(fr, wdw)
        monkeypatch.setenv("LSPIV_BLUR_ONE_COLUMN", "1")
        one = filters.smooth(fr, wdw)
        assert np.array_equal(got.view(np.uint32), one.view(np.uint32)), ("smooth", wdw)
        assert np.abs(got - fo.smooth(fr, wdw)).max() <= 2e-6 * 255
        assert np.array_equal(got[2], np.full(shape, 255.0, np.float32))   # the taps sum to one, exactly
    for w1, w2 in ((1, 2), (1, 3), (2, 3)):
        monkeypatch.delenv("LSPIV_BLUR_ONE_COLUMN", raising=False)
        got = filters.edge_detect(fr, w1, w2)
        monkeypatch.setenv("LSPIV_BLUR_ONE_COLUMN", "1")
        one = filters.edge_detect(fr, w1, w2)
        assert np.array_equal(got.view(np.uint32), one.view(np.uint32)), ("edge_detect", w1, w2)
        assert np.abs(got - fo.edge_detect(fr, w1, w2)).max() <= 4e-6 * 255


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
def test_gpu_register_ring_blur_of_run_time_radii_equals_the_lds_ring_kernel(gpu, monkeypatch, dtype):
    """Round 6: windows above 3 (k >= 9; the reference's user guide filters with wdw 2 | 4 and 6 | 10) go through blur_stripr_kernel -- the
    streaming strip kernel with the ring of row-filtered rows in REGISTERS (the row loop unrolled by the ring's period, laid out for
    RMAX = 5 / 10 -- windows 4 .. 10; 11 .. 15 stay with the LDS ring --, the taps beyond the run-time radius skipped by uniform branches).  The same expression per pixel as the LDS-ring
    kernel (LSPIV_BLUR_RING=1): the same bits for every radius class, on frames smaller than a strip and than the halo
    (BORDER_REFLECT_101 wraps), heights around the 64-row strips, NaN samples (a NaN outside a kernel's support must not reach the
    sum); and within the oracle's tolerance."""
    from pyorc_amd import filters

    rng = np.random.default_rng(5)
    for shape in ((3, 70, 150), (2, 33, 64), (2, 7, 9), (1, 130, 257)):
        fr = (rng.random(shape) * 256).astype(np.uint8)
        if dtype != np.uint8:
            fr = (fr.astype(dtype) - 90.25) * 0.5
            fr[0, shape[1] // 2, shape[2] // 2] = np.nan
        scale = 255.0
        for wdw in (4, 5, 7, 10, 12, 15):
            monkeypatch.delenv("LSPIV_BLUR_RING", raising=False)
            got = filters.smooth(fr, wdw)
            monkeypatch.setenv("LSPIV_BLUR_RING", "1")
            ring = filters.smooth(fr, wdw)
            assert np.array_equal(got.view(np.uint32), ring.view(np.uint32)), ("smooth", shape, wdw)
            ref = fo.smooth(np.nan_to_num(fr), wdw) if dtype == np.uint8 else None
            if ref is not None:
                assert np.abs(got - ref).max() <= 4e-6 * scale
        for w1, w2 in ((2, 4), (6, 10), (1, 5), (3, 8), (11, 15), (4, 4)):
            monkeypatch.delenv("LSPIV_BLUR_RING", raising=False)
            got = filters.edge_detect(fr, w1, w2)
            monkeypatch.setenv("LSPIV_BLUR_RING", "1")
            ring = filters.edge_detect(fr, w1, w2)
            assert np.array_equal(got.view(np.uint32), ring.view(np.uint32)), ("edge_detect", shape, w1, w2)
            if dtype == np.uint8:
                assert np.abs(got - fo.edge_detect(fr, w1, w2)).max() <= 8e-6 * scale
    monkeypatch.delenv("LSPIV_BLUR_RING", raising=False)
