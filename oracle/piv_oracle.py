"""CPU oracle for the LSPIV hot path (numpy, float64).  TEST INFRASTRUCTURE ONLY.

** PARITY UNPINNED ** -- read before trusting.

What this is
------------
A restatement, in plain numpy, of the algorithm behind pyorc's ``Frames.get_piv()``:

  pyorc/api/frames.py:114-197          parameter resolution  (``resolve_piv_args``)
  pyorc/velocimetry/ffpiv.py:24-179    chunk planning, 1-frame halo  (``plan_chunks``)
  pyorc/velocimetry/ffpiv.py:379-443   per-timestep loop + px -> m/s  (``get_ffpiv`` timestep branch)
  pyorc/velocimetry/ffpiv.py:446-474   corr_max / s2n / u,v  (``get_uv_timestep``)
  pyorc/velocimetry/ffpiv.py:182-376   ensemble correlation  (``get_ffpiv`` ensemble branch)

The arithmetic itself (window extraction, normalised FFT cross-correlation, Gaussian
sub-pixel peak) is NOT in /root/reference: it lives in the third-party PyPI package
``ffpiv >= 0.2.1`` (pyproject.toml:19; README.md:185 pins 0.2.1) on top of ``rocket_fft``
(pyproject.toml:38), neither of which is importable in the build container (no index
access).  Those parts are restated here from ffpiv's published algorithm (which mirrors
OpenPIV's ``pyprocess``): ``ffpiv.window.get_axis_shape/get_axis_coords/
get_rect_coordinates``, ``ffpiv.pivnp.normalize_intensity/ncc/multi_img_ncc/
peak_position/u_v_displacement``.  Every such function below is tagged [FFPIV-RESTATED].

Why "parity unpinned"
---------------------
The only numeric known-answer test of this path in the reference
(tests/test_frames.py:139-153) needs examples/ngwerere/ngwerere_20191103.mp4, which is
absent (.MISSING_LARGE_BLOBS:1), plus cv2/xarray/ffpiv, none of which are importable.
So this oracle could be checked against *no* golden vector of the reference and against
*no* output of the reference run here.  It is pinned only by (i) call-site contracts in
pyorc/velocimetry/ffpiv.py (shapes, dtypes, NaN conventions), (ii) analytic known
answers (integer / fractional shifts of synthetic images), and (iii) the int16x0.01
on-disk encoding of examples/ngwerere/ngwerere_piv.nc (pyorc/const.py:80).  Open
semantic questions are listed as A1..A8 in SURVEY.md section 8c and repeated at the function
that embodies each choice.

Who may use this file
---------------------
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- as the checker / reported baseline, never as the product path.  The
product (``pyorc_amd``) must never import it.

All arithmetic is float64 (numpy promotes uint8 - float64 mean -> float64, which is
what ffpiv does for uint8 stacks).  ``numpy.fft`` is pocketfft, the same FFT family
rocket_fft wraps.
"""

from __future__ import annotations

import warnings
from typing import Optional, Sequence, Tuple

import numpy as np

EPS_PEAK = 1e-7  # [FFPIV-RESTATED] eps added to the plane before the log-Gaussian fit (A5)

# The readings of ffpiv that nothing in /root/reference can decide.  The defaults are
# the oracle's choice; the alternatives exist here AND in the HIP library (lspiv_set_option with the same names and
# values) so that whichever a real ffpiv turns out to do is a one-line default flip, each covered by a GPU test:
#   border_peak      0: arg-max on the plane border -> (nan, nan) (OpenPIV's vectorised routine); 1: -> the plane
#                    centre, i.e. zero displacement (OpenPIV's scalar find_subpixel_peak_position returns
#                    default_peak_position there); 2: -> the integer peak, no sub-pixel fit
#   signal_mode      0: signal_threshold scores each window PAIR, both windows must reach the fraction (CHANGELOG.md:57-60
#                    "any of the 2 interrogation window in a window pair"); 1: one score per window POSITION over the whole
#                    chunk ("fraction of non-zero pixels in the window stack", pyorc/velocimetry/ffpiv.py:93-97)
#   signal_positive  0: the score counts samples != 0 ("non-zero pixels"); 1: samples > 0 ("intensities above zero")
# Round 3 widened the set (VERDICT r02 "what's missing" 2): every other reading the restatement hard-coded is a switch too,
# so that a future ffpiv run can only end in "combination X matches":
#   v_sign           0: v = row shift of the peak as it comes out of the plane (positive = down the image rows; inferred from
#                    pyorc/api/plot.py:548,576-583, which flips v only for plotting); 1: v negated inside the engine
#   norm_clip        1: negative lobes of the normalised window are removed, clip(x, 0, max) (A3; OpenPIV's normalize_intensity);
#                    0: no clip, plain (a - mean) / std
#   std_ddof         0: population standard deviation (numpy's default, A3); 1: sample standard deviation (n - 1)
#   round_odd        round_to_even for odd window sizes (A8): 0 round-half-even of x / 2 (25 -> 24, 27 -> 28); 1 up to the next
#                    even number (25 -> 26); 2 down (27 -> 26)
SEMANTICS = {"border_peak": 0, "signal_mode": 0, "signal_positive": 0, "v_sign": 0, "norm_clip": 1, "std_ddof": 0, "round_odd": 0}


class semantics:
    """``with semantics(border_peak=1): ...`` -- evaluate the oracle under an alternative reading."""

    def __init__(self, **kw):
        unknown = set(kw) - set(SEMANTICS)
        if unknown:
            raise KeyError(f"unknown semantics {sorted(unknown)}")
        self.kw = kw

    def __enter__(self):
        self.saved = dict(SEMANTICS)
        SEMANTICS.update(self.kw)
        return self

    def __exit__(self, *exc):
        SEMANTICS.clear()
        SEMANTICS.update(self.saved)


# ----------------------------------------------------------------------------------------------
# ffpiv.window  [FFPIV-RESTATED]
# ----------------------------------------------------------------------------------------------
def round_to_even(input_tuple: Sequence[float]) -> Tuple[int, ...]:
    """Round window sizes to even integers (call site pyorc/api/frames.py:167).

    A8 (unverified): direction for odd sizes.  Restated as round-half-even of x/2 times 2
    (25 -> 24, 27 -> 28); irrelevant for the even sizes of every BASELINE config.
    """
    mode = SEMANTICS["round_odd"]
    if mode == 1:
        return tuple(int(np.ceil(float(x) / 2.0) * 2) for x in input_tuple)
    if mode == 2:
        return tuple(int(np.floor(float(x) / 2.0) * 2) for x in input_tuple)
    return tuple(int(np.round(float(x) / 2.0) * 2) for x in input_tuple)


def get_axis_shape(dim_size: int, window_size: int, overlap: int) -> int:
    """Number of interrogation windows along one axis (A1): (dim - win)//(win - overlap) + 1."""
    if window_size <= overlap:
        raise ValueError("overlap must be smaller than window size")
    if dim_size < window_size:
        return 0
    return int((dim_size - window_size) // (window_size - overlap) + 1)


def get_axis_coords(dim_size: int, window_size: int, overlap: int) -> np.ndarray:
    """Integer pixel centres of the windows along one axis (A2).

    arange(n)*(win - overlap) + win/2 cast to int64; windows are anchored top-left
    (``center_on_field=False``), the remainder pixels at bottom/right are unused.
    """
    n = get_axis_shape(dim_size, window_size, overlap)
    coords = np.arange(n) * (window_size - overlap) + window_size / 2.0
    return np.int64(coords)


def get_rect_coordinates(dim_size, window_size, overlap, search_area_size=None):
    """Window-centre coordinates ``(x_cols, y_rows)`` (call site pyorc/api/frames.py:85-90)."""
    search_area_size = window_size if search_area_size is None else search_area_size
    y = get_axis_coords(dim_size[0], search_area_size[0], overlap[0])
    x = get_axis_coords(dim_size[1], search_area_size[1], overlap[1])
    return x, y


def window_origins(dim_size, window_size, overlap):
    """Top-left pixel of every window along y and x (centre - win//2)."""
    x, y = get_rect_coordinates(dim_size, window_size, overlap)
    return y - window_size[0] // 2, x - window_size[1] // 2


def sliding_window_stack(imgs: np.ndarray, window_size, overlap) -> np.ndarray:
    """Materialise the window stack (T, n_win, wy, wx), window index row-major k*n_cols+m.

    Row-major order is implied by the reshape at pyorc/velocimetry/ffpiv.py:469.
    """
    imgs = np.asarray(imgs)
    T, H, W = imgs.shape
    wy, wx = window_size
    y0, x0 = window_origins((H, W), window_size, overlap)
    iy = y0[:, None] + np.arange(wy)[None, :]  # (n_rows, wy)
    ix = x0[:, None] + np.arange(wx)[None, :]  # (n_cols, wx)
    # (T, n_rows, n_cols, wy, wx)
    stack = imgs[:, iy[:, None, :, None], ix[None, :, None, :]]
    return stack.reshape(T, len(y0) * len(x0), wy, wx)


def required_memory(n_frames, dim_size, window_size, overlap, search_area_size=None) -> float:
    """Bytes ffpiv materialises for a stack (call site pyorc/velocimetry/ffpiv.py:120-126).

    Window stack + correlation volume in float64 (restated; only feeds the chunk planner).
    """
    n_rows = get_axis_shape(dim_size[0], window_size[0], overlap[0])
    n_cols = get_axis_shape(dim_size[1], window_size[1], overlap[1])
    per_frame = n_rows * n_cols * window_size[0] * window_size[1] * 8.0
    return float(n_frames * per_frame * 2.0)


# ----------------------------------------------------------------------------------------------
# ffpiv.pivnp  [FFPIV-RESTATED]
# ----------------------------------------------------------------------------------------------
def normalize_intensity(win: np.ndarray) -> np.ndarray:
    """Per-window normalisation inside ncc (A3): (a-mean)/std (0 if std==0), clipped to >= 0."""
    win = np.asarray(win, dtype=np.float64)
    off = win - win.mean(axis=(-2, -1), keepdims=True)
    std = off.std(axis=(-2, -1), keepdims=True, ddof=SEMANTICS["std_ddof"])
    out = np.divide(off, std, out=np.zeros_like(off), where=(std != 0))
    # clip(x, 0, max(x)): the upper bound never binds, the lower removes negative lobes
    return np.maximum(out, 0.0) if SEMANTICS["norm_clip"] else out


def ncc(win_a: np.ndarray, win_b: np.ndarray) -> np.ndarray:
    """Normalised circular cross-correlation of window pairs (A4).

    clip(fftshift(irfft2(conj(rfft2 a) * rfft2 b)) / (wy*wx), 0, 1)
    """
    wy, wx = win_a.shape[-2:]
    a = normalize_intensity(win_a)
    b = normalize_intensity(win_b)
    fa = np.conj(np.fft.rfft2(a))
    fb = np.fft.rfft2(b)
    c = np.fft.irfft2(fa * fb, s=(wy, wx))
    c = np.fft.fftshift(c, axes=(-2, -1)) / float(wy * wx)
    return np.clip(c, 0.0, 1.0)


def signal_mask(stack_a: np.ndarray, stack_b: np.ndarray, threshold: Optional[float]) -> np.ndarray:
    """True where a window PAIR carries enough signal (A7).

    fraction of non-zero pixels, evaluated per window of the pair; the pair is kept only if
    both windows reach the threshold (CHANGELOG.md:57-60, docs/user-guide/velocimetry/
    index.rst:90-94).  "non-zero" vs "> 0" is unverified; identical for non-negative frames.
    """
    if threshold is None:
        return np.ones(stack_a.shape[:-2], dtype=bool)
    n = stack_a.shape[-1] * stack_a.shape[-2]
    fa = _count_signal(stack_a) / n
    fb = _count_signal(stack_b) / n
    return (fa >= threshold) & (fb >= threshold)


def _count_signal(stack: np.ndarray) -> np.ndarray:
    """Samples that count as signal per window: != 0, or > 0 under SEMANTICS["signal_positive"]."""
    return ((stack > 0) if SEMANTICS["signal_positive"] else (stack != 0)).sum(axis=(-2, -1))


def signal_mask_stack(stack: np.ndarray, threshold: Optional[float]) -> np.ndarray:
    """SEMANTICS["signal_mode"] == 1: one flag per window POSITION, the fraction of signal samples of that position over
    all frames of the chunk (float32 score like the HIP kernel: count / (T * wy * wx), compared in float32)."""
    if threshold is None:
        return np.ones(stack.shape[1], dtype=bool)
    T, _, wy, wx = stack.shape
    score = (_count_signal(stack).sum(axis=0).astype(np.float64) / (float(T) * wy * wx)).astype(np.float32)
    return score >= np.float32(threshold)


def cross_corr(imgs, window_size=(64, 64), overlap=(32, 32), search_area_size=None,
               normalize=False, engine="numpy", signal_threshold=None, verbose=False):
    """ffpiv.cross_corr restated: returns (x, y, corr) with corr (T-1, n_win, wy, wx) float64.

    Call sites: pyorc/velocimetry/ffpiv.py:222-231, 450-459.  ``normalize`` (stack-level
    normalisation) is always False from pyorc and not implemented.  Planes of window pairs
    below ``signal_threshold`` are NaN (pyorc/velocimetry/ffpiv.py:93-97).
    """
    if normalize:
        raise NotImplementedError("stack-level normalize is never used by pyorc (ffpiv.py:227,455)")
    imgs = np.asarray(imgs)
    if imgs.ndim != 3 or imgs.shape[0] < 2:
        raise ValueError("imgs must be (T>=2, H, W)")
    search_area_size = window_size if search_area_size is None else search_area_size
    if tuple(search_area_size) != tuple(window_size):
        raise NotImplementedError("pyorc always passes search_area_size == window_size (frames.py:168)")
    x, y = get_rect_coordinates(imgs.shape[-2:], window_size, overlap)
    stack = sliding_window_stack(imgs, window_size, overlap)
    T = imgs.shape[0]
    corr = np.full((T - 1,) + stack.shape[1:], np.nan, dtype=np.float64)
    keep_pos = signal_mask_stack(stack, signal_threshold) if SEMANTICS["signal_mode"] == 1 else None
    for t in range(T - 1):
        keep = signal_mask(stack[t], stack[t + 1], signal_threshold) if keep_pos is None else keep_pos
        if keep.any():
            corr[t, keep] = ncc(stack[t, keep], stack[t + 1, keep])
    return x, y, corr


def peak_position(plane: np.ndarray) -> Tuple[float, float]:
    """Sub-pixel peak (row, col) of one correlation plane (A5).

    flat argmax (first maximum in row-major order; NaN counts as maximum like np.argmax);
    peak on the border -> (nan, nan); otherwise 3-point log-Gaussian fit on plane + 1e-7 in
    both directions, zero denominator -> zero offset.
    """
    wy, wx = plane.shape
    idx = int(np.argmax(plane))
    i, j = idx // wx, idx % wx
    if i == 0 or i == wy - 1 or j == 0 or j == wx - 1:
        mode = 0 if np.isnan(plane[i, j]) else SEMANTICS["border_peak"]   # a NaN plane stays NaN in every mode
        return (np.nan, np.nan) if mode == 0 else (float(wy // 2), float(wx // 2)) if mode == 1 else (float(i), float(j))
    c = plane[i, j] + EPS_PEAK
    cl = plane[i - 1, j] + EPS_PEAK
    cr = plane[i + 1, j] + EPS_PEAK
    cd = plane[i, j - 1] + EPS_PEAK
    cu = plane[i, j + 1] + EPS_PEAK
    with np.errstate(all="ignore"):
        nom1 = np.log(cl) - np.log(cr)
        den1 = 2 * np.log(cl) - 4 * np.log(c) + 2 * np.log(cr)
        nom2 = np.log(cd) - np.log(cu)
        den2 = 2 * np.log(cd) - 4 * np.log(c) + 2 * np.log(cu)
    di = nom1 / den1 if den1 != 0.0 else 0.0
    dj = nom2 / den2 if den2 != 0.0 else 0.0
    return i + di, j + dj


def u_v_displacement(corr: np.ndarray, n_rows: int, n_cols: int, engine: str = "numpy", eps: float = None):
    """ffpiv.u_v_displacement restated (call sites pyorc/velocimetry/ffpiv.py:324, 471).

    corr (P, n_win, wy, wx) -> u, v (P, n_rows, n_cols) in pixels: u = column shift,
    v = row shift, relative to the plane centre floor(w/2); no sign flip
    (pyorc/api/plot.py:548,576-583).  ``eps``: what is added to the plane before the logarithms (default EPS_PEAK = 1e-7,
    the constant the kernels use too; the keyword exists for tests/golden/regen_from_ffpiv.py, which finds out from a
    real ffpiv's planes and displacements which value it uses).
    """
    eps = EPS_PEAK if eps is None else float(eps)
    corr = np.asarray(corr)
    if corr.ndim == 3:
        corr = corr[None]
    P, n_win, wy, wx = corr.shape
    assert n_win == n_rows * n_cols
    u = np.full((P, n_rows, n_cols), np.nan)
    v = np.full((P, n_rows, n_cols), np.nan)
    ci, cj = wy // 2, wx // 2
    # vectorised argmax + gather (equivalent to peak_position per plane)
    flat = corr.reshape(P * n_win, wy * wx)
    idx = np.argmax(flat, axis=1)
    i, j = idx // wx, idx % wx
    ok = (i > 0) & (i < wy - 1) & (j > 0) & (j < wx - 1)
    r = np.nonzero(ok)[0]
    pl = corr.reshape(P * n_win, wy, wx)
    ii, jj = i[r], j[r]
    c = pl[r, ii, jj] + eps
    cl = pl[r, ii - 1, jj] + eps
    cr = pl[r, ii + 1, jj] + eps
    cd = pl[r, ii, jj - 1] + eps
    cu = pl[r, ii, jj + 1] + eps
    with np.errstate(all="ignore"):
        lc, lcl, lcr, lcd, lcu = np.log(c), np.log(cl), np.log(cr), np.log(cd), np.log(cu)
        nom1, den1 = lcl - lcr, 2 * lcl - 4 * lc + 2 * lcr
        nom2, den2 = lcd - lcu, 2 * lcd - 4 * lc + 2 * lcu
        di = np.divide(nom1, den1, out=np.zeros_like(nom1), where=(den1 != 0.0))
        dj = np.divide(nom2, den2, out=np.zeros_like(nom2), where=(den2 != 0.0))
    uf = np.full(P * n_win, np.nan)
    vf = np.full(P * n_win, np.nan)
    uf[r] = jj + dj - cj
    vf[r] = ii + di - ci
    edge = ~ok & ~np.isnan(flat[np.arange(flat.shape[0]), idx])   # a NaN plane (skipped window) stays NaN in every mode
    if SEMANTICS["border_peak"] == 1:      # plane centre: zero displacement
        uf[edge] = 0.0
        vf[edge] = 0.0
    elif SEMANTICS["border_peak"] == 2:    # integer peak
        uf[edge] = (j - cj)[edge]
        vf[edge] = (i - ci)[edge]
    if SEMANTICS["v_sign"]:
        vf = -vf
    return uf.reshape(P, n_rows, n_cols), vf.reshape(P, n_rows, n_cols)


# ----------------------------------------------------------------------------------------------
# pyorc/velocimetry/ffpiv.py restated (numpy in, dict out -- xarray is not importable here)
# ----------------------------------------------------------------------------------------------
def get_uv_timestep(imgs, n_cols, n_rows, window_size, overlap, signal_threshold=None):
    """pyorc/velocimetry/ffpiv.py:446-474: u, v [px], corr_max, s2n (float32) per pair."""
    _, _, corr = cross_corr(imgs, window_size=window_size, overlap=overlap, signal_threshold=signal_threshold)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        corr_max = np.nanmax(corr, axis=(-1, -2))
        s2n = corr_max / np.nanmean(corr, axis=(-1, -2))
    s2n = s2n.reshape(-1, n_rows, n_cols).astype(np.float32)
    corr_max = corr_max.reshape(-1, n_rows, n_cols).astype(np.float32)
    u, v = u_v_displacement(corr, n_rows, n_cols)
    return u, v, corr_max, s2n


def plan_chunks(n_frames: int, req_mem: float, avail_mem: float, chunksize: Optional[int] = None):
    """Chunk planner of pyorc/velocimetry/ffpiv.py:127-142 -> list of (start, stop) frame slices.

    ``avail_mem`` is already divided by memory_factor.  Quirk Q1 (user chunksize -> NameError
    in the reference, ffpiv.py:140) is fixed as chunks = ceil(T/chunksize).
    """
    if chunksize is None:
        chunks = int((req_mem // avail_mem) + 1)
        chunksize = int(np.ceil(n_frames / chunks))
        if chunksize <= 5:
            chunksize = 5
            chunks = int(np.ceil(n_frames / chunksize))
    else:
        chunks = int(np.ceil(n_frames / chunksize))
    if chunksize < 2:
        raise OverflowError(f"Chunk size with selected nr of chunks ({chunks}) is 2 or less.")
    slices = [(max(k * chunksize - 1, 0), min((k + 1) * chunksize, n_frames)) for k in range(chunks)]
    return [(a, b) for a, b in slices if b - a >= 2]


def get_ffpiv(frames, dt, window_size, overlap, res_y, res_x, chunksize=None, avail_mem=None,
              ensemble_corr=False, corr_min=0.2, s2n_min=3.0, count_min=0.2, signal_threshold=None):
    """pyorc/velocimetry/ffpiv.py:24-179 restated on plain arrays.

    frames (T,H,W); dt (T-1,) seconds per pair.  Returns dict(s2n, corr, v_x, v_y) of float32
    arrays (P, n_rows, n_cols) with P = T-1 (timestep) or 1 (ensemble); plus "pair_index":
    index of the 2nd frame of every pair (the reference labels results with da.time[1:]).
    """
    frames = np.asarray(frames)
    dt = np.asarray(dt, dtype=np.float64)
    T, H, W = frames.shape
    n_rows = get_axis_shape(H, window_size[0], overlap[0])
    n_cols = get_axis_shape(W, window_size[1], overlap[1])
    req = required_memory(T, (H, W), window_size, overlap)
    avail = float("inf") if avail_mem is None else float(avail_mem)
    chunks = plan_chunks(T, req, avail, chunksize)
    if not ensemble_corr:
        out = {"s2n": [], "corr": [], "v_x": [], "v_y": [], "pair_index": []}
        for a, b in chunks:
            u, v, corr_max, s2n = get_uv_timestep(frames[a:b], n_cols, n_rows, window_size, overlap,
                                                  signal_threshold)
            dtc = dt[a:b - 1][:, None, None]  # dt.sel(time=da.time[1:]) -> pair index = 2nd frame - 1
            out["v_x"].append((u * res_x / dtc).astype(np.float32))
            out["v_y"].append((v * res_y / dtc).astype(np.float32))
            out["corr"].append(corr_max)
            out["s2n"].append(s2n)
            out["pair_index"].append(np.arange(a + 1, b))
        return {k: np.concatenate(vv, axis=0) for k, vv in out.items()}
    # ensemble branch, pyorc/velocimetry/ffpiv.py:182-376
    corr_sum, corr_count = 0.0, 0.0
    cm_chunks, s2n_chunks = [], []
    for a, b in chunks:
        _, _, corr = cross_corr(frames[a:b], window_size=window_size, overlap=overlap,
                                signal_threshold=signal_threshold)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            corr_max = np.max(corr, axis=(-1, -2))
            s2n = corr_max / np.mean(corr, axis=(-1, -2))
        masks = (corr_max >= corr_min) & (s2n >= s2n_min) & np.isfinite(corr_max)
        corr[~masks] = 0.0
        corr_max[~masks] = 0.0
        s2n[~masks] = 0.0
        corr_sum = corr_sum + np.sum(corr, axis=0, keepdims=True)
        corr_count = corr_count + np.sum(corr_max > 1e-6, axis=0, keepdims=True)
        cm_chunks.append(corr_max)
        s2n_chunks.append(s2n)
    n_frames = len(cm_chunks)  # quirk Q3: number of CHUNKS, not pairs (ffpiv.py:373)
    s2n_concat = np.concatenate(s2n_chunks, axis=0)
    cm_concat = np.concatenate(cm_chunks, axis=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        low = corr_count < count_min * n_frames
        corr_sum = np.array(corr_sum, dtype=np.float64)
        corr_sum[low] = np.nan
        cm_concat[:, low.flatten()] = np.nan
        with np.errstate(all="ignore"):
            corr_mean = np.divide(corr_sum, corr_count[..., None, None])
        corr_max_mean = np.nanmean(cm_concat, axis=0).reshape(-1, n_rows, n_cols)
        s2n_mean = np.nanmean(s2n_concat, axis=0).reshape(-1, n_rows, n_cols)
    u, v = u_v_displacement(corr_mean, n_rows, n_cols)
    dt_av = dt.mean()
    return {
        "s2n": s2n_mean, "corr": corr_max_mean,
        "v_x": (u * res_x / dt_av).astype(np.float32), "v_y": (v * res_y / dt_av).astype(np.float32),
        "pair_index": np.array([chunks[-1][0] + 1]),  # quirk Q3: time[0:1] of the LAST chunk (ffpiv.py:336)
        "corr_mean": corr_mean,
    }


def resolve_piv_args(window_size, overlap=None):
    """pyorc/api/frames.py:159-171: window -> (wy,wx) even; search area = window; default overlap."""
    ws_in = window_size
    ws = 2 * (ws_in,) if isinstance(ws_in, (int, np.integer)) else tuple(ws_in)
    ws_even = round_to_even(ws)
    if overlap is None:
        if isinstance(ws_in, (int, np.integer)):
            overlap = 2 * (int(round(ws_in) / 2),)  # quirk Q6: from the un-rounded size
        else:
            overlap = tuple(int(round(w) / 2) for w in ws_in)
    return ws_even, ws_even, tuple(overlap)


def encode_int16(a: np.ndarray, scale=0.01, fill=-9999) -> np.ndarray:
    """On-disk packing of pyorc/const.py:80 (int16, scale 0.01, fill -9999) with the arithmetic xarray's CF encoder
    applies to float32 variables: float32 data / float32(scale), NaN -> fill, np.around (half to even), int16."""
    a = np.asarray(a, dtype=np.float32)
    q = a / np.float32(scale)
    q = np.where(np.isnan(q), np.float32(fill), np.around(q))
    return np.clip(q, -32768, 32767).astype(np.int16)
