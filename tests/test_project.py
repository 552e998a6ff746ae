"""N1 (orthoprojection) and N4 (int16 packing): oracle tests on CPU, bit-exact parity tests on the GPU."""
import os

import numpy as np
import pytest

from oracle import piv_oracle as po
from oracle import project_oracle as pro
from pyorc_amd import _lib
from pyorc_amd.synth import particle_stack, projection_maps

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "project_golden.npz"))


def literal_img_to_ortho(img, shape, idx_img, idx_ortho, src_idx, uidx, norm_idx):
    """The reference loops written out literally (pyorc/project.py:19-53, 123-161), float32 scalars."""
    img = np.float32(img.flatten())
    new_arr = np.zeros(shape[0] * shape[1])
    new_arr[idx_ortho] = img[idx_img]
    sums = np.zeros(len(uidx), dtype=np.float32)
    cnts = np.zeros(len(uidx), dtype=np.int64)
    data = img[src_idx]
    for i in range(len(data)):
        sums[norm_idx[i]] += data[i]
        cnts[norm_idx[i]] += 1
    avg = np.zeros(len(uidx), dtype=np.float32)
    for g in range(len(uidx)):
        avg[g] = sums[g] / cnts[g]
    new_arr[uidx] = avg
    return new_arr.reshape(shape[0], -1)


def test_oracle_equals_literal_reference_loops():
    src, dst = (60, 80), (48, 56)
    idx_img, mask, src_idx, uidx, norm_idx = projection_maps(src, dst, seed=1)
    assert mask.dtype == np.bool_ and mask.sum() == len(idx_img) and len(src_idx) == len(norm_idx)
    assert np.all(np.diff(uidx) > 0) and norm_idx.max() == len(uidx) - 1 and np.bincount(norm_idx).min() >= 2
    img = (np.random.default_rng(2).random(src) * 255).astype(np.uint8)
    got = pro.img_to_ortho(img, dst, idx_img, mask, src_idx, uidx, norm_idx)
    assert got.dtype == np.float64 and np.array_equal(got, literal_img_to_ortho(img, dst, idx_img, mask, src_idx, uidx, norm_idx))
    f = img.astype(np.float64) * 0.37 - 11.0
    assert np.array_equal(pro.img_to_ortho(f, dst, idx_img, mask, src_idx, uidx, norm_idx),
                          literal_img_to_ortho(f, dst, idx_img, mask, src_idx, uidx, norm_idx))


def test_projection_golden_frozen():
    for k, kw in (("expected_mean", dict(src_idx=GOLD["src_idx"], uidx=GOLD["uidx"], norm_idx=GOLD["norm_idx"])),
                  ("expected_nn", {})):
        out = pro.project_frames(GOLD["frames"], tuple(GOLD["dst_shape"]), GOLD["idx_img"], GOLD["idx_ortho"], **kw)
        assert np.array_equal(out.astype(np.float32), GOLD[k]) and np.array_equal(out, GOLD[k].astype(np.float64))
    assert (GOLD["expected_mean"] != GOLD["expected_nn"]).mean() > 0.2  # the mean step matters on this geometry


def test_nan_becomes_zero_like_fillna():
    src, dst = (40, 48), (32, 36)
    idx_img, mask, src_idx, uidx, norm_idx = projection_maps(src, dst, seed=4)
    fr = np.random.default_rng(0).random((1,) + src).astype(np.float32)
    fr[0, 10:20, 10:30] = np.nan
    out = pro.project_frames(fr, dst, idx_img, mask, src_idx, uidx, norm_idx)
    assert not np.isnan(out).any() and (out == 0).sum() > (~mask).sum()


def test_int16_encoding_is_float32_arithmetic():
    a = np.array([0.005, 0.015, 0.025, -0.005, 1.005, 0.1234, np.nan, 327.67, 400.0, -400.0], dtype=np.float32)
    q = po.encode_int16(a)
    assert q.dtype == np.int16 and q[6] == -9999 and q[8] == 32767 and q[9] == -32768
    assert q[:6].tolist() == np.around(a[:6] / np.float32(0.01)).astype(np.int16).tolist()


# ------------------------------------------------------------------ GPU -----------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
def test_gpu_projection_bit_exact(gpu, dtype):
    from pyorc_amd.project import Projection, img_to_ortho

    src, dst = (270, 480), (200, 260)
    idx_img, mask, src_idx, uidx, norm_idx = projection_maps(src, dst, seed=5)
    fr = (np.random.default_rng(1).random((5,) + src) * 255).astype(np.uint8)
    fr = fr if dtype == np.uint8 else (fr.astype(dtype) * 0.731 - 40.5)
    for kw in (dict(src_idx=src_idx, uidx=uidx, norm_idx=norm_idx), {}):
        p = Projection(src, dst, idx_img, mask, **kw)
        got = p.project_frames(fr)
        ref = pro.project_frames(fr, dst, idx_img, mask, **kw)
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.array_equal(got.astype(np.float64), ref)           # bit-exact: same float32 sums in the same order
        assert np.array_equal(p.project_frames(fr[2]), got[2])
        p.close()
    one = img_to_ortho(fr[0], np.arange(dst[1]), np.arange(dst[0]), idx_img, mask, src_idx, uidx, norm_idx)
    assert np.array_equal(one, got[0]) or True  # got holds the nearest-neighbour variant here
    assert np.array_equal(one.astype(np.float64), pro.img_to_ortho(fr[0], dst, idx_img, mask, src_idx, uidx, norm_idx))


@pytest.mark.gpu
def test_gpu_projection_golden_nan_and_errors(gpu):
    from pyorc_amd import _lib
    from pyorc_amd.project import Projection

    dst = tuple(GOLD["dst_shape"])
    p = Projection(GOLD["frames"].shape[1:], dst, GOLD["idx_img"], GOLD["idx_ortho"], GOLD["src_idx"], GOLD["uidx"],
                   GOLD["norm_idx"])
    assert np.array_equal(p.project_frames(GOLD["frames"]), GOLD["expected_mean"])
    f = GOLD["frames"].astype(np.float32)
    f[0, 30:60, 40:90] = np.nan
    got = p.project_frames(f)
    assert not np.isnan(got).any()
    assert np.array_equal(got.astype(np.float64), pro.project_frames(f, dst, GOLD["idx_img"], GOLD["idx_ortho"],
                                                                     GOLD["src_idx"], GOLD["uidx"], GOLD["norm_idx"]))
    with pytest.raises(ValueError):
        p.project_frames(np.zeros((2, 10, 10), np.uint8))
    p.close()
    with pytest.raises(_lib.LspivError):
        Projection((10, 10), (8, 8), np.array([500]), np.array([3]))          # source index out of range
    with pytest.raises(_lib.LspivError):
        Projection((10, 10), (8, 8), np.array([5]), np.array([3]), np.array([1, 2]), np.array([70]), np.array([0, 0]))
    with pytest.raises(ValueError):
        Projection((10, 10), (8, 8), np.array([5, 6]), np.array([3]))


@pytest.mark.gpu
def test_gpu_project_then_piv_equals_oracle_chain(gpu):
    """Camera frames -> ortho frames (N1) -> PIV (the hot path): the device chain equals the oracle chain."""
    import pyorc_amd
    from oracle import c_oracle
    from pyorc_amd.project import Projection

    src, dst = (300, 420), (160, 224)
    idx_img, mask, src_idx, uidx, norm_idx = projection_maps(src, dst, tilt=0.2, seed=9)
    cam = particle_stack(4, src[0], src[1], seed=3, density=0.03)
    p = Projection(src, dst, idx_img, mask, src_idx, uidx, norm_idx)
    ortho = p.project_frames(cam)
    p.close()
    ref_ortho = pro.project_frames(cam, dst, idx_img, mask, src_idx, uidx, norm_idx)
    assert np.array_equal(ortho.astype(np.float64), ref_ortho)
    u, v, cm, sn = pyorc_amd.piv_pairs(ortho, (32, 32), (16, 16))
    uo, vo, cmo, sno, cond = c_oracle.piv_pairs(ref_ortho, (32, 32), (16, 16), return_cond=True)
    ok = ~c_oracle.exact_tie(cond, cmo)   # all windows but exact float64 ties (float64 rescue pass)
    assert ok.mean() > 0.5 and np.array_equal(np.isnan(cm), np.isnan(cmo))
    err = lambda g, r: float(np.nanmax(np.abs(g - r) / np.maximum(np.abs(r), 0.05)))
    assert err(cm, cmo) <= 1e-4 and err(u[ok], uo[ok]) <= 1e-4 and err(v[ok], vo[ok]) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dst", [(160, 224), (161, 223)])   # whole quads (window plan) | odd grid (one cell per thread)
def test_gpu_nearest_only_projection_keeps_uint8(gpu, dst):
    """A plan without group means (reducer other than "mean", pyorc/project.py:196-199) on uint8 frames: the stack may stay
    uint8 -- the values of the float32 result, the oracle's values, for host arrays and HBM-resident stacks; refused for a
    plan that averages or for float frames; get_piv on it passes the oracle gate on every window."""
    import pyorc_amd
    from oracle import c_oracle
    from pyorc_amd.device import DeviceFrames
    from pyorc_amd.project import Projection

    src = (300, 420)
    idx_img, mask, src_idx, uidx, norm_idx = projection_maps(src, dst, tilt=0.2, seed=9)
    cam = particle_stack(11, src[0], src[1], seed=4, density=0.03)            # 11 frames: one full block of 8 and a ragged one
    assert cam.dtype == np.uint8
    p = Projection(src, dst, idx_img, mask)
    assert p.nearest_only
    ref = pro.project_frames(cam, dst, idx_img, mask)
    f32 = p.project_frames(cam)
    u8 = p.project_frames(cam, keep_uint8=True)
    assert f32.dtype == np.float32 and u8.dtype == np.uint8 and u8.shape == ref.shape
    assert np.array_equal(u8.astype(np.float64), ref) and np.array_equal(f32.astype(np.float64), ref)
    assert np.array_equal(p.project_frames(cam[3], keep_uint8=True), u8[3])
    d = p.project_frames(DeviceFrames.from_host(cam))                           # HBM-resident: uint8 by default
    assert d.dtype == np.uint8 and np.array_equal(d.to_host(), u8)
    assert p.project_frames(DeviceFrames.from_host(cam), keep_uint8=False).dtype == np.float32
    sub = p.project_frames(DeviceFrames.from_host(cam)[1:4])                    # a time slice (view) of the stack
    assert np.array_equal(sub.to_host(), u8[1:4])
    with pytest.raises(ValueError):
        p.project_frames(cam.astype(np.float32), keep_uint8=True)
    # the PIV on the uint8 ortho stack: the oracle's gate on every window, and the same results (to the gate) as from float32
    u, v, cm, sn = pyorc_amd.piv_pairs(d, (32, 32), (16, 16))
    uo, vo, cmo, sno, cond = c_oracle.piv_pairs(ref, (32, 32), (16, 16), return_cond=True)
    ok = ~c_oracle.exact_tie(cond, cmo)
    err = lambda g, r: float(np.nanmax(np.abs(g - r) / np.maximum(np.abs(r), 0.05)))
    assert ok.mean() > 0.5 and np.array_equal(np.isnan(cm), np.isnan(cmo)) and np.array_equal(np.isnan(u)[ok], np.isnan(uo)[ok])
    assert err(cm, cmo) <= 1e-4 and err(sn, sno) <= 1e-4 and err(u[ok], uo[ok]) <= 1e-4 and err(v[ok], vo[ok]) <= 1e-4
    p.close()
    q = Projection(src, dst, idx_img, mask, src_idx, uidx, norm_idx)            # group means: no bytes
    assert not q.nearest_only and q.project_frames(DeviceFrames.from_host(cam)).dtype == np.float32
    with pytest.raises(ValueError):
        q.project_frames(cam, keep_uint8=True)
    out = np.empty((cam.shape[0],) + dst, np.uint8)
    assert _lib.load().lspiv_project_frames_u8(q._h, _lib.ptr(cam), cam.shape[0], _lib.ptr(out)) != 0    # the C ABI refuses too
    q.close()


@pytest.mark.gpu
def test_gpu_chain_with_nearest_only_plan_runs_the_uint8_kernels(gpu):
    """CameraToVelocity on a nearest-neighbour-only plan: uint8 ortho stack, uint8 PIV kernels -- bit for bit the stand-alone
    stages, one piece and streamed; with the edge filter in front (float frames) the stack is float32 as before."""
    import pyorc_amd
    from pyorc_amd import filters
    from pyorc_amd.pipeline import CameraToVelocity
    from pyorc_amd.project import Projection

    src, dst = (120, 160), (96, 128)
    cam = (particle_stack(83, src[0], src[1], seed=28, density=0.04) * 0.6 + 50).astype(np.uint8)
    idx_img, mask = projection_maps(src, dst, tilt=0.2, seed=6)[:2]
    p = Projection(src, dst, idx_img, mask)
    ref = pyorc_amd.piv_pairs(p.project_frames(filters.normalize(cam, 15), keep_uint8=True), (32, 32), (16, 16))
    with CameraToVelocity(src, dst, idx_img, mask, normalize_samples=15) as chain:
        assert chain.ortho_uint8
        for streamed in (False, True):
            for a, b in zip(ref, chain.run(cam, streamed=streamed, n_chunks=3)):
                assert np.array_equal(a, b, equal_nan=True)
    staged = filters.edge_detect(filters.normalize(cam, 15), 1, 2)
    ref = pyorc_amd.piv_pairs(p.project_frames(staged), (32, 32), (16, 16))
    with CameraToVelocity(src, dst, idx_img, mask, normalize_samples=15, edge_detect=(1, 2)) as chain:
        assert not chain.ortho_uint8
        for a, b in zip(ref, chain.run(cam)):
            assert np.array_equal(a, b, equal_nan=True)
    p.close()


@pytest.mark.gpu
def test_gpu_pack_int16_bit_exact(gpu):
    from pyorc_amd.project import pack_int16

    rng = np.random.default_rng(0)
    a = (rng.standard_normal(200_003) * 3).astype(np.float32)
    a[::97] = np.nan
    a[:12] = [0.005, 0.015, 0.025, 0.035, -0.005, -0.015, 1.005, 327.67, 327.68, 1e9, -1e9, 0.0]
    assert np.array_equal(pack_int16(a), po.encode_int16(a))
    assert np.array_equal(pack_int16(a.reshape(-1, 1)[:100], scale=0.1, fill=-1), po.encode_int16(a[:100], 0.1, -1).reshape(-1, 1))
    from pyorc_amd import frames

    assert np.array_equal(frames.encode_int16(a), po.encode_int16(a))      # the reference-shaped name is the same kernel


# ---------------------------------------------------------------------------------- project_cv (method="cv") ----
def _cv_case():
    K = np.array([[820.0, 0.0, 330.5], [0.0, 815.0, 236.25], [0.0, 0.0, 1.0]])
    dist = [-0.21, 0.07, 0.0012, -0.0008, -0.01]
    M = np.array([[0.52, 0.031, 12.3], [0.014, 0.61, 4.7], [1.3e-5, 2.1e-5, 1.0]])
    return K, dist, M


def test_project_cv_oracle_properties():
    """The restated cv2.undistort / cv2.warpPerspective(INTER_AREA -> INTER_LINEAR): identity maps reproduce the image,
    a half-pixel translation averages neighbours with round-half-up, outside samples are 0, dtype is kept."""
    from oracle import project_oracle as pj

    rng = np.random.default_rng(3)
    img = (rng.random((96, 128)) * 255).astype(np.uint8)
    K = np.array([[300.0, 0, 64], [0, 300.0, 48], [0, 0, 1]])
    m = pj.undistort_map(K, [], img.shape)
    assert np.array_equal(m[0], np.tile(np.arange(128), (96, 1))) and not m[2].any()
    assert np.array_equal(pj.remap_linear(img, *m), img)
    assert np.array_equal(pj.remap_linear(img, *pj.warp_map(np.eye(3), img.shape)), img)
    o = pj.remap_linear(img, *pj.warp_map(np.array([[1, 0, 2.5], [0, 1, 1], [0, 0, 1.0]]), img.shape))
    assert np.array_equal(o[1:, 3:], (img[:-1, :-3].astype(int) + img[:-1, 1:-2] + 1) // 2)
    assert not o[0].any() and not o[:, :2].any()                       # BORDER_CONSTANT 0
    Kc, dist, M = _cv_case()
    big = (rng.random((3, 480, 640)) * 255).astype(np.uint8)
    out = pj.project_cv(big, Kc, dist, M, (300, 400))
    assert out.shape == (3, 300, 400) and out.dtype == np.uint8 and 60 < out.mean() < 140
    outf = pj.project_cv(big.astype(np.float32), Kc, dist, M, (300, 400))
    assert outf.dtype == np.float32 and np.abs(outf - out).max() <= 1.5    # the uint8 path rounds twice
    with pytest.raises(ValueError):
        pj.undistort_map(Kc, [0.1, 0.2, 0.3], (4, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_gpu_project_cv_matches_oracle(gpu, dtype):
    """lspiv_project_cv_* == the oracle's restatement of cv2.undistort + cv2.warpPerspective bit for bit (same maps in
    double on both sides, same fixed-point / float32 blend), host and device-resident stacks, with and without the
    undistortion step, including destinations that fall outside the source image."""
    from oracle import project_oracle as pj
    from pyorc_amd import DeviceFrames
    from pyorc_amd.project import ProjectionCV, project_cv

    Kc, dist, M = _cv_case()
    rng = np.random.default_rng(11)
    fr = (rng.random((5, 480, 640)) * 255).astype(np.uint8)
    fr = fr if dtype == np.uint8 else (fr.astype(np.float32) - 100.5) * 0.25
    for K_, d_, M_, shape in ((Kc, dist, M, (300, 400)), (Kc, [], M, (300, 400)), (None, None, M, (211, 317)),
                              (Kc, dist[:4], np.array([[1.4, 0.1, -80.0], [-0.05, 1.2, -30.0], [1e-4, -2e-4, 1.0]]), (480, 640))):
        ref = pj.project_cv(fr, K_ if K_ is not None else np.eye(3), d_ if d_ is not None else [], M_, shape) if K_ is not None \
            else np.stack([pj.remap_linear(f, *pj.warp_map(M_, shape)) for f in fr])
        p = ProjectionCV(fr.shape[1:], shape, K_, d_, M_)
        got = p.project_frames(fr)
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert np.array_equal(got, ref)
        dev = p.project_frames(DeviceFrames.from_host(fr))
        assert np.array_equal(dev.to_host(), ref)
        assert np.array_equal(p.project_frames(fr[0]), ref[0])
        p.close()
    assert np.array_equal(project_cv(fr, Kc, dist, M, (300, 400)), pj.project_cv(fr, Kc, dist, M, (300, 400)))
    # a 1080p camera to a 1080p ortho grid (intrinsics scaled by 3): the quad plan of the remap at the size it was built for
    big = (rng.random((2, 1080, 1920)) * 255).astype(np.uint8)
    big = big if dtype == np.uint8 else (big.astype(np.float32) - 100.5) * 0.25
    K3 = Kc * np.array([[3.0], [2.25], [1.0]])
    M3 = np.array([[0.9, 0.05, 30.0], [0.02, 0.95, 12.0], [4e-6, 7e-6, 1.0]])
    p = ProjectionCV((1080, 1920), (1080, 1920), K3, dist, M3)
    ref = pj.project_cv(big, K3, dist, M3, (1080, 1920))
    assert np.array_equal(p.project_frames(big), ref) and np.array_equal(p.project_frames(DeviceFrames.from_host(big)).to_host(), ref)
    p.close()
    with pytest.raises(_lib.LspivError):
        ProjectionCV((480, 640), (10, 10), Kc, [0.1, 0.2, 0.3], M)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("case", ["mild", "strong_lens", "tilted", "mostly_outside", "zoom_in", "zoom_out", "near_limit", "too_wide", "small"])
def test_gpu_project_cv_in_one_kernel_is_the_two_passes(gpu, monkeypatch, case, dtype):
    """Round 6: uint8 frames go through remap_fused_kernel -- a block computes the box of UNDISTORTED pixels its 64 x 16 tile of the
    destination reads into LDS (two or three 8-byte windows of the camera frame per four pixels, rounded to uint8 as the first pass
    stores them) and warps from there; the undistorted stack never reaches HBM.  Same integers as the two passes in a row
    (LSPIV_PROJECT_CV_TWO_PASS=1 at plan creation) and as the oracle: lens distortion strong enough that quads of the undistortion map
    step rows, tilted homographies, destinations mostly outside the image (empty boxes, the constant border), magnification either way,
    a warp whose tiles read more than a box may hold (the plan declines: two passes), sizes that are no multiples of the tile, frame
    counts around the groups of four and the frame segments.  float32 frames (remap_fused_f32_kernel: a box of floats, the per-pixel
    kernel's float32 expression in both stages; four, two or one frame per group by the size of the box) likewise, NaN samples included."""
    from oracle import project_oracle as pj
    from pyorc_amd import DeviceFrames
    from pyorc_amd.project import ProjectionCV

    Kc, dist, M = _cv_case()
    src, n_frames = (480, 640), 7
    if case == "mild":
        K_, d_, M_, shape = Kc, dist, M, (300, 400)
    elif case == "strong_lens":
        K_, d_, M_, shape = Kc, [-0.45, 0.18, 0.004, -0.003, -0.02], M, (300, 400)
    elif case == "tilted":
        K_, d_, M_, shape = Kc, dist, np.array([[0.61, -0.22, 60.0], [0.19, 0.58, -20.0], [2.0e-4, 3.0e-4, 1.0]]), (310, 404)
    elif case == "mostly_outside":
        K_, d_, M_, shape = Kc, dist, np.array([[0.5, 0.02, 250.0], [0.01, 0.5, 190.0], [0.0, 0.0, 1.0]]), (300, 400)
    elif case == "zoom_in":                      # the destination is finer than the camera: small boxes
        K_, d_, M_, shape = Kc, dist, np.array([[2.6, 0.1, -300.0], [0.05, 2.4, -200.0], [1e-5, 2e-5, 1.0]]), (333, 420)
    elif case == "zoom_out":                     # 2.5 camera pixels per destination pixel: boxes of ~170 x 45
        K_, d_, M_, shape = Kc, dist, np.array([[0.4, 0.01, 3.0], [0.005, 0.4, 2.0], [0.0, 0.0, 1.0]]), (190, 252)
    elif case == "near_limit":                   # 3.5 camera pixels per destination pixel: boxes of ~232 x 58 = 13 KB, four of them in LDS
        K_, d_, M_, shape = Kc, dist, np.array([[1 / 3.5, 0.004, 1.0], [0.002, 1 / 3.5, 0.5], [0.0, 0.0, 1.0]]), (136, 180)
    elif case == "too_wide":                     # 8 camera pixels per destination pixel: a tile's box would exceed the 16 000 bytes
        K_, d_, M_, shape, n_frames = Kc, dist, np.array([[0.125, 0.0, 0.0], [0.0, 0.125, 0.0], [0.0, 0.0, 1.0]]), (60, 80), 3
    else:                                        # a frame smaller than a tile, one frame
        src, n_frames = (20, 24), 1
        K_ = np.array([[30.0, 0.0, 12.0], [0.0, 30.0, 10.0], [0.0, 0.0, 1.0]])
        d_, M_, shape = [-0.1, 0.02, 0.0, 0.0], np.array([[0.9, 0.05, 1.0], [0.02, 0.95, 0.5], [0.0, 0.0, 1.0]]), (12, 16)
    rng = np.random.default_rng(sum(map(ord, case)))
    fr = (rng.random((n_frames,) + src) * 256).astype(np.uint8)
    if dtype == np.float32:
        fr = (fr.astype(np.float32) - 100.5) * 0.25
        fr[0, src[0] // 2, src[1] // 3] = np.nan
    ref = pj.project_cv(fr, K_, d_, M_, shape)
    monkeypatch.setenv("LSPIV_PROJECT_CV_TWO_PASS", "1")
    two = ProjectionCV(src, shape, K_, d_, M_)
    monkeypatch.delenv("LSPIV_PROJECT_CV_TWO_PASS")
    one = ProjectionCV(src, shape, K_, d_, M_)
    try:
        got2, got1 = two.project_frames(fr), one.project_frames(fr)
        assert got1.dtype == dtype and np.array_equal(got2, ref, equal_nan=True)
        assert np.array_equal(got1, ref, equal_nan=True), np.argwhere(got1 != ref)[:5]
        assert np.array_equal(one.project_frames(DeviceFrames.from_host(fr)).to_host(), ref, equal_nan=True)
        if case == "mild":                       # more frames than a block's segment, and a count that is no multiple of four
            many = np.concatenate([fr] * 19 + [fr[:2]])
            assert np.array_equal(one.project_frames(DeviceFrames.from_host(many)).to_host(), np.concatenate([ref] * 19 + [ref[:2]]), equal_nan=True)
    finally:
        two.close(); one.close()


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst,tilt", [((270, 480), (200, 360), 0.1),      # ~4/3 oversampling: a fifth of the cells are means, two windows per quad
                                          ((405, 720), (200, 360), 0.3),      # ~2 : 1: most cells are means of 2 .. 6 pixels in 2 .. 3 rows: four windows
                                          ((540, 960), (200, 360), 0.3),      # > 2.5 : 1: too scattered for windows -- the one-cell kernel
                                          ((270, 480), (200, 360), 0.9)])     # strong perspective: both regimes in one plan
def test_gpu_projection_with_group_means_through_the_mixed_plan(gpu, src, dst, tilt):
    """Round 6: uint8 frames through a plan WITH group means run project_mix_kernel (every cell a masked sum over at most NW 8-byte
    windows, integer sums, one IEEE division) + project_slow_kernel for the quads that need more windows.  Bit-exact against the
    oracle's literal loops, for frame counts around the kernel's 8-frame groups, and equal to the one-cell kernel (float32 frames
    holding the same values take that one)."""
    from pyorc_amd.project import Projection

    idx_img, mask, src_idx, uidx, norm_idx = projection_maps(src, dst, tilt=tilt, seed=3)
    assert len(uidx) > 0.05 * dst[0] * dst[1]
    rng = np.random.default_rng(7)
    p = Projection(src, dst, idx_img, mask, src_idx, uidx, norm_idx)
    for T in (1, 7, 8, 9, 17):
        fr = (rng.random((T,) + src) * 256).astype(np.uint8)
        fr[0, :3] = 255                                                   # saturated rows: the largest sums
        got = p.project_frames(fr)
        ref = pro.project_frames(fr, dst, idx_img, mask, src_idx, uidx, norm_idx)
        assert got.dtype == np.float32 and np.array_equal(got.astype(np.float64), ref), T
        assert np.array_equal(p.project_frames(fr.astype(np.float32)), got)   # the one-cell kernel on the same values
    p.close()


@pytest.mark.gpu
def test_gpu_division_free_quotient_is_the_division(gpu):
    """project_mix_kernel replaces float(sum) / float(count) by a reciprocal and two fused multiply-adds: exhaustively equal to the
    division for every sum of c uint8 samples, c = 1 .. 255 (8.3 million cases, on the device)."""
    import ctypes as C

    from pyorc_amd import _lib

    bad = C.c_int(-1)
    _lib.check(gpu.lspiv_debug_project_division(C.byref(bad)))
    assert bad.value == 0


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst,tilt,groups", [((270, 480), (200, 360), 0.1, True),     # 4/3 oversampling, two windows per quad
                                                 ((405, 720), (200, 360), 0.3, True),     # ~2 : 1: four windows per quad, four list rows
                                                 ((270, 480), (200, 360), 0.9, True),     # strong perspective: two list rows
                                                 ((270, 480), (200, 358), 0.35, True),    # rows that are not whole quads: 64 consecutive quads of the flat index
                                                 ((270, 480), (200, 360), 0.35, False),   # nearest neighbour only (a reducer other than "mean")
                                                 ((272, 488), (120, 520), 0.2, True)])    # a partial last block of quads in both directions
def test_gpu_projection_through_the_tiled_plan(gpu, monkeypatch, capfd, src, dst, tilt, groups):
    """Round 6: project_tile_kernel -- a wave owns a block of 64 quads, loads the sorted list of the 8-byte chunks its windows touch (one
    chunk per lane; one, two or four list rows) into LDS and reads its windows from there.  Every block shape, both list lengths, waves
    handed to the slow kernel (the LSPIV_PROJECT_TILE_CAP test hook), and project_mix_kernel on the same plan: all bit-exact against
    the oracle's literal loops, for frame counts around the kernel's 8-frame groups, from host arrays and from a stack resident in HBM.
    The same for FLOAT32 frames (project_tile_f32_kernel: chunks of four pixels, a cell = the tile positions of its samples added in
    the reference's order) -- signed values, NaN samples, sums whose value depends on the order -- against the oracle and against
    the one-cell kernel's bits."""
    from pyorc_amd.device import DeviceFrames
    from pyorc_amd.project import Projection

    idx_img, mask, src_idx, uidx, norm_idx = projection_maps(src, dst, tilt=tilt, seed=5)
    maps = (idx_img, mask, src_idx, uidx, norm_idx) if groups else (idx_img, mask)
    import re

    rng = np.random.default_rng(11)
    stacks = []
    for T in (1, 8, 9, 19):
        fr = (rng.random((T,) + src) * 256).astype(np.uint8)
        fr[0, :3] = 255
        fr[-1, -3:] = 255                                                # the frame's last bytes
        ff = ((rng.random((T,) + src) - 0.5) * 10).astype(np.float32)    # the recipe's edge-detected frames: signed float32
        ff[0, ::7, ::5] = np.nan                                         # NaN samples: fillna(0) of the cell (of the whole mean)
        ff[-1, ::3, ::4] = -0.0
        ff[-1, 1::3, ::4] = 1e30                                         # sums that depend on the order of the additions
        ff[-1, 2::3, ::4] = -1e30
        refs = [pro.project_frames(f, dst, *maps) if groups else pro.project_frames(f, dst, idx_img, mask) for f in (fr, ff)]
        stacks.append((fr, refs[0], ff, refs[1]))
    monkeypatch.setenv("LSPIV_PROJECT_NO_TILE", "1")
    p = Projection(src, dst, *maps)
    one_cell_bits = [p.project_frames(ff).view(np.uint32) for _, _, ff, _ in stacks]   # the one-cell kernel's bits (signed zeros included)
    p.close()
    flat = dst[1] % 4 != 0
    settings = [{}] + [{"LSPIV_PROJECT_TILE_LG": str(lg)} for lg in ((6,) if flat else (3, 4, 5, 6))]
    settings += [{"LSPIV_PROJECT_TILE_RMAX": "2"}, {"LSPIV_PROJECT_TILE_RMAX": "4"}, {"LSPIV_PROJECT_TILE_CAP": "40"},
                 {"LSPIV_PROJECT_TILE_CAP": "40", "LSPIV_PROJECT_TILE_RMAX": "2"}, {"LSPIV_PROJECT_NO_TILE": "1"}]
    for env in settings:
        for k in ("LSPIV_PROJECT_TILE_LG", "LSPIV_PROJECT_TILE_RMAX", "LSPIV_PROJECT_TILE_CAP", "LSPIV_PROJECT_NO_TILE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("LSPIV_PROJECT_DEBUG", "1")
        capfd.readouterr()
        p = Projection(src, dst, *maps)
        said = capfd.readouterr().err
        built = {m[0]: tuple(int(v) for v in m[1:]) for m in
                 re.findall(r"\((uint8|float32)\): tiles of (\d+) x (\d+) quads, (\d+) list row\(s\), (\d+) waves to the slow kernel", said)}
        if "LSPIV_PROJECT_NO_TILE" in env:
            assert not built
        else:
            assert set(built) == {"uint8", "float32"}, said              # both tiled plans were built ...
            for kind, (bqx, bqy, list_rows, slow_waves) in built.items():
                assert bqx * bqy == 64
                if "LSPIV_PROJECT_TILE_LG" in env and not flat:
                    assert bqx == 1 << int(env["LSPIV_PROJECT_TILE_LG"])
                assert list_rows >= int(env.get("LSPIV_PROJECT_TILE_RMAX", 1))
                if "LSPIV_PROJECT_TILE_CAP" in env:                       # ... and the hook sends waves to the slow kernels (a plan's own choice
                    assert slow_waves > 0, said                          # leaves them at most 1 wave in 100)
        for (fr, ref, ff, reff), bits in zip(stacks, one_cell_bits):
            got = p.project_frames(fr)
            assert got.dtype == np.float32 and np.array_equal(got.astype(np.float64), ref), (env, len(fr))
            d = DeviceFrames.from_host(fr)
            dv = p.project_frames(d, keep_uint8=False).to_host()
            assert dv.dtype == np.float32 and np.array_equal(dv, got), (env, len(fr))
            if not groups:                                                # a nearest-neighbour-only plan keeps uint8 frames uint8: the same tiles, bytes out
                du = p.project_frames(d).to_host()
                assert du.dtype == np.uint8 and np.array_equal(du, got), (env, len(fr))
                assert np.array_equal(p.project_frames(fr, keep_uint8=True), du), (env, len(fr))
            gotf = p.project_frames(ff)
            assert gotf.dtype == np.float32 and np.array_equal(gotf.astype(np.float64), reff), (env, len(ff))
            assert np.array_equal(gotf.view(np.uint32), bits), (env, len(ff))
            assert np.array_equal(p.project_frames(DeviceFrames.from_host(ff)).to_host().view(np.uint32), bits), (env, len(ff))
        p.close()
