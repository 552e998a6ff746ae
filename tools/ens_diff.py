"""Numerical difference of two tools/ens_hash.py dumps (ENS_DUMP=a.npz / b.npz): two builds of the ensemble kernels on the same stack."""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    nan_same = np.array_equal(np.isnan(x), np.isnan(y))
    d = np.abs(x - y); ref = np.maximum(np.abs(x), 1e-30)
    rel = np.nanmax(d / np.maximum(np.abs(x), np.nanmax(np.abs(x)) * 1e-3)) if x.size else 0.0
    print(f"{k:9s} equal {np.array_equal(x, y, equal_nan=True)!s:5s} nan-mask-equal {nan_same!s:5s} differing {int((d > 0).sum())}/{x.size} max abs {np.nanmax(d):.3e} max rel (floor 1e-3 of max) {rel:.3e}")
