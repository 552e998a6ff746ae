"""Pin the oracle on a REAL ffpiv the moment one is importable.

    python tests/golden/regen_from_ffpiv.py            # report only
    python tests/golden/regen_from_ffpiv.py --write    # also write tests/golden/ffpiv_pinned.npz

Neither ``ffpiv`` (>= 0.2.1, pyproject.toml:19 of the reference) nor ``rocket_fft`` / ``numba`` can be imported in the
build container or on the GPU box, so every PIV parity claim of this repository is "vs own oracle" (DESIGN.md section 0).
This script is what closes that gap on a machine that has ffpiv: it feeds the inputs of the committed golden
fixtures (``piv_golden.npz``: G1 known shifts, G2 degenerate windows, G3 mini stack) to
``ffpiv.cross_corr`` / ``ffpiv.u_v_displacement`` exactly the way pyorc calls them
(pyorc/velocimetry/ffpiv.py:450-471: ``normalize=False``, ``search_area_size == window_size``, then
``nanmax`` / ``nanmean`` over the planes) and

  1. diffs the outputs against the oracle under every combination of the unpinned readings
     (``piv_oracle.SEMANTICS``: border_peak 0/1/2, signal_mode, signal_positive, v_sign, norm_clip, std_ddof -- 96 combinations;
     round_odd is checked on its own) and prints which combination matches -- the defaults of oracle AND HIP library
     (``lspiv_set_option``, same names and values) are then flipped to it;
  2. places what is left: the peak fit alone on ffpiv's OWN planes under nine values of the eps added before the logarithms
     (A5; a constant, not a switch -- the report names the four places to change), and the planes alone (A3 / A4);
     checks ``ffpiv.window.round_to_even`` on odd sizes (A8: 25 -> 24 or 26) and the grid functions (A1 / A2);
  3. with ``--write`` stores the ffpiv outputs as ``ffpiv_pinned.npz`` next to the inputs' names; when that file exists
     ``tests/test_oracle.py::test_oracle_matches_pinned_ffpiv_outputs`` and the GPU parity tests assert against it and
     the "parity unpinned" caveat can go.

Without ffpiv it exits with status 3 and says so (``tests/test_oracle.py`` checks that it does, so the script cannot rot).
The reference's own known answer (tests/test_frames.py:139-153, two 4-value ``v_x`` vectors on the Ngwerere clip)
additionally needs the absent mp4 + cv2 + xarray; ``--ngwerere FRAMES.npy`` accepts the projected frame stack of that
test (``frames_proj`` fixture, 3 x 475 x 371) saved from a full pyorc installation and checks those eight numbers.
"""

from __future__ import annotations

import argparse
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF_VX_ENSEMBLE = [0.10917795, 0.10898168, 0.11020568, 0.12450387]   # /root/reference/tests/test_frames.py:148-149
REF_VX_TIMESTEP = [0.10837663, 0.11250661, 0.11100861, 0.1231317]    # /root/reference/tests/test_frames.py:150-151


def ffpiv_outputs(ffpiv, frames, ws, ov, thr=None, engine="numba"):
    """What pyorc/velocimetry/ffpiv.py:446-474 computes from ffpiv's two entry points."""
    x, y, corr = ffpiv.cross_corr(frames, window_size=ws, overlap=ov, search_area_size=ws, normalize=False, engine=engine,
                                  signal_threshold=thr, verbose=False)
    n_rows, n_cols = len(y), len(x)
    with np.errstate(all="ignore"):
        corr_max = np.nanmax(corr, axis=(-1, -2))
        s2n = corr_max / np.nanmean(corr, axis=(-1, -2))
    u, v = ffpiv.u_v_displacement(corr.astype(np.float32), n_rows, n_cols, engine=engine)
    return dict(u=np.asarray(u, np.float64), v=np.asarray(v, np.float64),
                corr=corr_max.reshape(-1, n_rows, n_cols), s2n=s2n.reshape(-1, n_rows, n_cols), planes=corr)


def oracle_outputs(po, frames, ws, ov, thr=None):
    x, y, corr = po.cross_corr(frames, ws, ov, signal_threshold=thr)
    u, v, cm, sn = po.get_uv_timestep(frames, len(x), len(y), ws, ov, thr)
    return dict(u=u, v=v, corr=cm, s2n=sn, planes=corr)


def compare(a, b, tol=1e-4):
    """(same NaN masks, worst relative error over u, v, corr, s2n)"""
    worst, same = 0.0, True
    for k in ("u", "v", "corr", "s2n"):
        same = same and np.array_equal(np.isnan(a[k]), np.isnan(b[k]))
        with np.errstate(all="ignore"):
            e = np.abs(a[k] - b[k]) / np.maximum(np.abs(b[k]), 0.05)
        if np.isfinite(e).any():
            worst = max(worst, float(np.nanmax(e)))
    return same, worst


def cases(gold):
    for k in range(len(gold["g1_frames"])):
        yield f"g1_{k}", gold["g1_frames"][k], (32, 32), (16, 16), None
    for name, fr in zip(gold["g2_names"], gold["g2_frames"]):
        yield f"g2_{name}", fr, (32, 32), (16, 16), None
    for tag in ("u8", "f32"):
        yield f"g3_{tag}_32", gold[f"g3_frames_{tag}"], (32, 32), (16, 16), None
        yield f"g3_{tag}_64", gold[f"g3_frames_{tag}"], (64, 64), (48, 48), None
    yield "g3_u8_32_thr03", gold["g3_frames_u8"], (32, 32), (16, 16), 0.3
    yield "g3_f32_32_thr03", gold["g3_frames_f32"], (32, 32), (16, 16), 0.3


def crafted_planes() -> np.ndarray:
    """(1, 12, 16, 16) float32 correlation planes, zero but for a peak of 0.9 at (7, 8) and chosen neighbours -- several of
    them exactly 0, where the value of the eps added before the logarithms decides the sub-pixel offset (A5)."""
    up = (0.0, 0.3, 0.0, 0.10, 0.5, 0.0, 0.05, 0.2, 0.0, 0.6, 0.01, 0.0)
    down = (0.3, 0.0, 0.1, 0.00, 0.2, 0.0, 0.00, 0.7, 0.8, 0.0, 0.00, 0.4)
    left = (0.2, 0.0, 0.0, 0.40, 0.0, 0.3, 0.60, 0.0, 0.1, 0.0, 0.02, 0.0)
    right = (0.0, 0.1, 0.5, 0.00, 0.0, 0.0, 0.20, 0.3, 0.0, 0.7, 0.00, 0.001)
    p = np.zeros((1, len(up), 16, 16), np.float32)
    for k in range(len(up)):
        p[0, k, 7, 8] = 0.9
        p[0, k, 6, 8], p[0, k, 8, 8], p[0, k, 7, 7], p[0, k, 7, 9] = up[k], down[k], left[k], right[k]
    return p


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--engine", default="numba")
    ap.add_argument("--ngwerere", default=None, help=".npy of the projected Ngwerere frames (3 x 475 x 371), see docstring")
    a = ap.parse_args(argv)
    try:
        import ffpiv
    except Exception as exc:  # ImportError, or numba failing to initialise
        print(f"ffpiv is not importable here ({type(exc).__name__}: {exc}); the PIV oracle stays PARITY-UNPINNED. "
              "Run this script where `pip install ffpiv>=0.2.1` works.")
        return 3
    from oracle import piv_oracle as po

    gold = np.load(os.path.join(HERE, "piv_golden.npz"))
    pinned = {}
    # every reading that is a switch in oracle + library: 3 x 2 x 2 x 2 x 2 x 2 = 96 combinations (round_odd is checked on its own below)
    names = ("border_peak", "signal_mode", "signal_positive", "v_sign", "norm_clip", "std_ddof")
    combos = list(itertools.product((0, 1, 2), (0, 1), (0, 1), (0, 1), (1, 0), (0, 1)))
    default = (0, 0, 0, 0, 1, 0)
    score = {c: [0, 0.0] for c in combos}   # cases matched, worst error
    n_cases = 0
    for name, fr, ws, ov, thr in cases(gold):
        n_cases += 1
        ref = ffpiv_outputs(ffpiv, fr, ws, ov, thr, a.engine)
        for k in ("u", "v", "corr", "s2n"):
            pinned[f"{name}_{k}"] = np.asarray(ref[k], np.float32)
        for c in combos:
            with po.semantics(**dict(zip(names, c))):
                same, worst = compare(oracle_outputs(po, fr, ws, ov, thr), ref)
            score[c][0] += int(same and worst <= 1e-4)
            score[c][1] = max(score[c][1], worst if same else np.inf)
    print(f"ffpiv {getattr(ffpiv, '__version__', '?')}, engine {a.engine}: {n_cases} golden cases, {len(combos)} combinations of {names}")
    ranked = sorted(combos, key=lambda c: (-score[c][0], score[c][1]))
    print("combination " + " ".join(names) + " | cases matched (NaN masks equal and <= 1e-4) | worst rel err   -- best 12 and the default")
    for c in ranked[:12] + ([default] if default not in ranked[:12] else []):
        tag = "  <- oracle default" if c == default else ""
        print(f"   {c}   {score[c][0]:3d} / {n_cases}   {score[c][1]:.3e}{tag}")
    best = ranked[0]
    print("best reading: " + ", ".join(f"{k}={v}" for k, v in zip(names, best))
          + ("  (= the defaults)" if best == default and score[best][0] == n_cases else
             "  -> flip piv_oracle.SEMANTICS and the option defaults in pyorc_amd/csrc/lspiv_api.hip"))
    if score[best][0] < n_cases:
        print(f"NOTE: no combination matches all {n_cases} cases (best: {score[best][0]}); what is left unexplained is outside the switch set "
              "(eps of the peak fit, FFT normalisation, grid) -- the stage reports below place it.")
    # the stages on their own, on ffpiv's OWN planes -- so that a mismatch above can be placed:
    #   A5: the peak fit of the best reading applied to ffpiv's planes, under several values of the eps added before the logs
    #       (a constant, not a switch: EPS_PEAK in oracle/piv_oracle.py and piv_oracle.c, kEpsPeak in pyorc_amd/csrc/common.h,
    #       eps in csrc/piv_rescue.hip -- four places to change if this names another value);
    #   A3 / A4: the oracle's planes against ffpiv's planes (normalisation, FFT scale, clip), no peak fit involved
    eps_values = (1e-7, 0.0, 1e-12, 1e-10, 1e-9, 1e-8, 1e-6, 1e-5, 1e-4)
    eps_score = {e: [0, 0.0] for e in eps_values}
    plane_worst, plane_cases = 0.0, 0

    def score_eps(planes32, n_rows, n_cols, u_ref, v_ref):
        for e in eps_values:
            uo, vo = po.u_v_displacement(planes32.astype(np.float64), n_rows, n_cols, eps=e)
            same, worst = True, 0.0
            for g, r in ((uo, u_ref), (vo, v_ref)):
                same = same and np.array_equal(np.isnan(g), np.isnan(r))
                with np.errstate(all="ignore"):
                    err = np.abs(g - r) / np.maximum(np.abs(r), 0.05)
                if np.isfinite(err).any():
                    worst = max(worst, float(np.nanmax(err)))
            eps_score[e][0] += int(same and worst <= 1e-4)
            eps_score[e][1] = max(eps_score[e][1], worst if same else np.inf)

    with po.semantics(**dict(zip(names, best))):
        # planes made for the question: a peak whose neighbours are exactly zero on one or both sides -- there the offset is
        # (ln eps - ln c) / (2 ln eps - ...) and moves by per cents between 1e-7 and 1e-9; the golden windows hardly notice eps
        crafted = crafted_planes()
        uc, vc = ffpiv.u_v_displacement(crafted, 1, crafted.shape[1], engine=a.engine)
        score_eps(crafted, 1, crafted.shape[1], np.asarray(uc, np.float64).reshape(1, 1, -1), np.asarray(vc, np.float64).reshape(1, 1, -1))
        n_eps_cases = 1
        for name, fr, ws, ov, thr in cases(gold):
            ref = ffpiv_outputs(ffpiv, fr, ws, ov, thr, a.engine)
            n_rows, n_cols = ref["u"].shape[-2:]
            score_eps(np.asarray(ref["planes"]).astype(np.float32), n_rows, n_cols, ref["u"], ref["v"])   # the float32 planes ffpiv was given
            n_eps_cases += 1
            op = oracle_outputs(po, fr, ws, ov, thr)["planes"]
            if op.shape == np.asarray(ref["planes"]).shape and np.array_equal(np.isnan(op), np.isnan(ref["planes"])):
                plane_cases += 1
                plane_worst = max(plane_worst, float(np.nanmax(np.abs(op - ref["planes"]), initial=0.0)))
    best_eps = sorted(eps_values, key=lambda e: (-eps_score[e][0], eps_score[e][1]))[0]
    print("A5 on ffpiv's own planes, eps of the peak fit | cases matched | worst rel err of u, v")
    for e in eps_values:
        print(f"   eps {e:7.0e}   {eps_score[e][0]:3d} / {n_eps_cases}   {eps_score[e][1]:.3e}" + ("  <- oracle and kernels" if e == 1e-7 else ""))
    print(f"best eps: {best_eps:g}" + ("  (= the constant in use)" if best_eps == 1e-7 else
                                       "  -> change EPS_PEAK (oracle/piv_oracle.py, .c), kEpsPeak (csrc/common.h) and eps (csrc/piv_rescue.hip)"))
    print(f"A3 / A4 planes, oracle vs ffpiv under the best reading: {plane_cases} / {n_cases} cases with equal shape and NaN mask, "
          f"worst |difference| {plane_worst:.3e} (float32 planes: <= ~1e-6 is agreement)")
    # A8 / A1 / A2
    from oracle import piv_oracle as po2
    for odd in (25, 27, 33, 11):
        want = tuple(ffpiv.window.round_to_even((odd, odd)))
        modes = []
        for m in (0, 1, 2):
            with po2.semantics(round_odd=m):
                if po2.round_to_even((odd, odd)) == want:
                    modes.append(m)
        print(f"round_to_even({odd}): ffpiv {want}  oracle default {po2.round_to_even((odd, odd))}  matching round_odd values {modes}")
    for dim, ws, ov in (((1080, 1920), (32, 32), (16, 16)), ((475, 371), (10, 10), (5, 5)), ((785, 875), (24, 24), (12, 12))):
        xf, yf = ffpiv.window.get_rect_coordinates(dim_size=dim, window_size=ws, search_area_size=ws, overlap=ov)
        xo, yo = po2.get_rect_coordinates(dim, ws, ov)
        print(f"grid {dim} {ws}/{ov}: {'equal' if np.array_equal(xf, xo) and np.array_equal(yf, yo) else 'DIFFERENT'}")
    if a.ngwerere:
        fr = np.load(a.ngwerere)
        for mode, ref in (("timestep", REF_VX_TIMESTEP), ("ensemble", REF_VX_ENSEMBLE)):
            res = po2.get_ffpiv(fr, np.full(len(fr) - 1, 1 / 30.0), (10, 10), (5, 5), 0.01, 0.01, ensemble_corr=(mode == "ensemble"),
                                corr_min=0.0, s2n_min=0.0, count_min=0.0)
            with np.errstate(all="ignore"):
                got = np.nanmean(res["v_x"], axis=0).flatten()[-4:]
            print(f"ngwerere {mode}: oracle v_x[-4:] = {got}, reference test expects {ref} "
                  f"({'OK' if np.allclose(got, ref, rtol=1e-5) else 'differs (dt / resolution of the fixture?)'})")
    if a.write:
        out = os.path.join(HERE, "ffpiv_pinned.npz")
        np.savez_compressed(out, ffpiv_version=str(getattr(ffpiv, "__version__", "?")), **pinned)
        print(f"wrote {out}")
    pinned_ok = score[default][0] == n_cases and best_eps == 1e-7
    print("RESULT: " + ("the oracle's defaults reproduce this ffpiv on every case, the peak fit's eps included" if pinned_ok else
                        "the oracle's defaults do NOT reproduce this ffpiv yet -- see 'best reading' / 'best eps' above"))
    return 0 if pinned_ok else 1


if __name__ == "__main__":
    sys.exit(main())
