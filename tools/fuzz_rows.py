"""Randomised differential test of the rows around the hot path: filters, both projections, the
post-PIV masks and the int16 packing against their numpy oracles, over random shapes (odd widths, single frames, tiny
frames), dtypes and parameters.  Bit-exact except the Gaussian filters (4e-6 of the value range) and `angle` (atan2f).
usage: fuzz_rows.py <seed> <cases>        FUZZ_KINDS=project,normalize restricts the kinds drawn; FUZZ_W4=1: widths that are multiples of
four (the four-column blur, project_cv in one kernel); FUZZ_DIST=0.3: amplitude of the lens coefficients of project_cv (default 0.05); FUZZ_BIGWDW=1: filter windows up to 15"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import filters_oracle as fo, mask_oracle as mo, piv_oracle as po, project_oracle as pj
from pyorc_amd import filters, mask as pm
from pyorc_amd.project import Projection, ProjectionCV, pack_int16
from pyorc_amd.synth import particle_stack, projection_maps

warnings.simplefilter("ignore")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
eq = lambda a, b: a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
bad = 0
t_start = time.time()
for case in range(n_cases):
    kind = str(rng.choice(os.environ["FUZZ_KINDS"].split(",") if os.environ.get("FUZZ_KINDS") else
                          ["time_diff", "range", "minmax", "normalize", "reduce_rolling", "blur", "project", "project_cv", "masks", "pack"]))
    T = int(rng.integers(1, 6)) if rng.random() < 0.3 else int(rng.integers(6, 40))
    H = int(rng.integers(3, 40)) if rng.random() < 0.3 else int(rng.integers(40, 200))
    W = int(rng.integers(3, 40)) if rng.random() < 0.3 else int(rng.integers(40, 260))
    dtype = rng.choice([np.uint8, np.float32, np.float64])
    if os.environ.get("FUZZ_W4"):            # widths the four-column blur / the one-kernel project_cv take (multiples of four), uint8 more often
        W = max(W // 4 * 4, 4)
        dtype = np.uint8 if rng.random() < 0.6 else dtype
    note, ok = "", True
    try:
        if kind in ("time_diff", "range", "blur"):
            fr = particle_stack(max(T, 2), H, W, seed=int(rng.integers(1 << 30)))
            if dtype != np.uint8:
                fr = fr.astype(dtype) * float(rng.uniform(0.1, 2)) - float(rng.uniform(0, 40))
                if rng.random() < 0.5:
                    fr[rng.integers(fr.shape[0]), rng.integers(H), rng.integers(W)] = np.nan
            if kind == "time_diff":
                thres, ab = float(rng.choice([0.0, 2.5, -1.0, 10.0])), bool(rng.random() < 0.5)
                ok = eq(filters.time_diff(fr, thres, ab), fo.time_diff(fr, thres, ab)); note = f"thres {thres} abs {ab}"
            elif kind == "range":
                ok = eq(filters.range(fr), fo.time_range(fr))
            else:
                fr = np.nan_to_num(fr)
                big = bool(os.environ.get("FUZZ_BIGWDW"))              # windows up to 15 (the library's limit: k = 31): every radius class of the run-time-radius kernel
                k1 = int(rng.integers(0, 13 if big else 7)); k2 = int(rng.integers(k1, 16 if big else 9))
                if rng.random() < 0.5:
                    got, ref = filters.smooth(fr, k2), fo.smooth(fr, k2); note = f"smooth {k2}"
                else:
                    got, ref = filters.edge_detect(fr, k1, k2), fo.edge_detect(fr, k1, k2); note = f"edge {k1} {k2}"
                scale = max(1.0, float(np.abs(fr).max()))
                err = float(np.abs(got - ref).max()) / scale
                ok = got.shape == ref.shape and err <= 4e-6; note += f" err {err:.1e}"
        elif kind == "minmax":
            f = (rng.standard_normal((max(T, 1), H, W)) * 6).astype(np.float32)
            f[0, 0, 0] = np.nan
            lo, hi = rng.choice([-5.0, -np.inf, 0.0]), rng.choice([5.0, np.inf, 2.0])
            ok = eq(filters.minmax(f, lo, hi), fo.minmax(f, lo, hi)); note = f"{lo} {hi}"
        elif kind in ("normalize", "reduce_rolling"):
            fr = particle_stack(max(T, 2), H, W, seed=int(rng.integers(1 << 30)))
            if rng.random() < 0.3:
                fr[rng.integers(fr.shape[0])] = int(rng.integers(0, 255))       # a constant frame
            if rng.random() < 0.3:
                fr[:, : H // 2, : W // 2] = 0
            samples = int(rng.integers(1, fr.shape[0] + 1))
            if kind == "normalize":
                if round(fr.shape[0] / samples) == 0:
                    samples = fr.shape[0]
                ok = eq(filters.normalize(fr, samples), fo.normalize(fr, samples))
            else:
                ok = eq(filters.reduce_rolling(fr, samples), fo.reduce_rolling(fr, samples))
            note = f"samples {samples}"
        elif kind == "project":
            src = (max(H, 24), max(W, 24)); dst = (int(rng.integers(8, 120)), int(rng.integers(8, 160)))
            maps = projection_maps(src, dst, tilt=float(rng.uniform(0.05, 0.5)), seed=int(rng.integers(1000)))
            fr = (rng.random((max(T, 1),) + src) * 255).astype(np.uint8)
            fr = fr if dtype == np.uint8 else fr.astype(dtype) * 0.731 - 40.5
            if dtype != np.uint8 and rng.random() < 0.5:                         # NaN samples: fillna(0) of the cell (a group's whole mean)
                fr[rng.integers(fr.shape[0]), ::int(rng.integers(1, 9)), ::int(rng.integers(1, 9))] = np.nan
            full = rng.random() < 0.7
            args = maps if full else maps[:2]
            p = Projection(src, dst, *args)
            ok = np.array_equal(p.project_frames(fr).astype(np.float64), pj.project_frames(fr, dst, *args))
            p.close(); note = f"dst {dst} groups {full}"
        elif kind == "project_cv":
            src = (max(H, 32), max(W, 32)); dst = (int(rng.integers(8, 150)), int(rng.integers(8, 200)))
            if os.environ.get("FUZZ_W4"):
                dst = (dst[0], dst[1] // 4 * 4)
            K = np.array([[rng.uniform(300, 900), 0, src[1] / 2 + rng.uniform(-5, 5)], [0, rng.uniform(300, 900), src[0] / 2 + rng.uniform(-5, 5)], [0, 0, 1.0]])
            nd = int(rng.choice([0, 4, 5, 8]))
            dist = list(rng.uniform(-1, 1, nd) * float(os.environ.get("FUZZ_DIST", 0.05)) * ([1, 1, 0.02, 0.02, 1, 1, 1, 1][:nd] if nd else []))
            M = np.array([[rng.uniform(0.4, 1.5), rng.uniform(-0.1, 0.1), rng.uniform(-20, 20)], [rng.uniform(-0.1, 0.1), rng.uniform(0.4, 1.5), rng.uniform(-20, 20)],
                          [rng.uniform(-2e-4, 2e-4), rng.uniform(-2e-4, 2e-4), 1.0]])
            d2 = np.uint8 if dtype == np.uint8 else np.float32
            fr = (rng.random((max(T, 1),) + src) * 255).astype(np.uint8)
            fr = fr if d2 == np.uint8 else (fr.astype(np.float32) - 100.5) * 0.25
            p = ProjectionCV(src, dst, K, dist, M)
            ok = eq(p.project_frames(fr), pj.project_cv(fr, K, dist, M, dst)); p.close(); note = f"dst {dst} dist {nd}"
        elif kind == "masks":
            R, Cc = int(rng.integers(1, 20)), int(rng.integers(1, 30))
            Tm = max(T, 2)
            f = np.empty((4, Tm, R, Cc), np.float32)
            f[0] = rng.normal(0.6, 0.5, (Tm, R, Cc)); f[1] = rng.normal(-0.1, 0.3, (Tm, R, Cc)); f[2] = rng.random((Tm, R, Cc)); f[3] = rng.random((Tm, R, Cc)) * 30
            f[:, rng.random((Tm, R, Cc)) < rng.uniform(0, 0.4)] = np.nan
            w = int(rng.integers(1, 4)); tol = float(rng.uniform(0.1, 0.9)); md = int(rng.integers(0, 2))
            cases = [("minmax", dict(s_min=0.1, s_max=tol * 3), [0.1, tol * 3]), ("count", dict(tolerance=tol), [tol]), ("corr", dict(tolerance=tol), [tol]),
                     ("s2n", dict(tolerance=tol * 20), [tol * 20]), ("outliers", dict(tolerance=tol * 2, mode="and" if md else "or"), [tol * 2, md]),
                     ("variance", dict(tolerance=tol * 5, mode="and" if md else "or"), [tol * 5, md]), ("rolling", dict(wdw=w + 1, tolerance=tol), [w + 1, tol]),
                     ("window_nan", dict(tolerance=tol, wdw=w), [tol, -w, w, -w, w]), ("window_mean", dict(tolerance=tol, wdw=w, mode="and" if md else "or"), [tol, md, -w, w, -w, w])]
            for name, kw, params in cases:
                got, ref = pm.run_mask(f, name, params), getattr(mo, name)(f, **kw)
                if not (got.shape == ref.shape and np.array_equal(got, ref)):
                    ok = False; note += f"{name} {kw} differs in {int((got != ref).sum())}; "
            ok = ok and eq(pm.time_mean(f), mo.time_mean(f))
            rep = pm.run_mask  # window_replace goes through the Mask wrapper in the tests; the kernels above cover its stencil
        else:
            a = (rng.standard_normal((max(T, 1), H, W)) * float(rng.choice([0.01, 1.0, 200.0]))).astype(np.float32)
            a[rng.random(a.shape) < 0.1] = np.nan
            ok = eq(pack_int16(a), po.encode_int16(a))
    except Exception as e:   # an error of the library on a valid case is a failure too
        ok, note = False, f"{type(e).__name__}: {e}"
    bad += not ok
    print(f"{'ok  ' if ok else 'FAIL'} {case:3d} {kind:14s} ({T},{H},{W}) {np.dtype(dtype).name:7s} {note}", flush=True)
print(f"{n_cases} cases, {bad} failures, {time.time()-t_start:.1f} s")
sys.exit(1 if bad else 0)
