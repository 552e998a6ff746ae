#!/usr/bin/env python
"""Kernel time of one PIV configuration on an HBM-resident synthetic stack, HIP events on the launch stream.

    [LSPIV_LIBRARY=other.so] [LSPIV_RESCUE=0] python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 [--dtype f32] [--tag name]

Prints one line: tag, ms per launch (mean of --reps after --warm untimed launches), pairs/s, rescue counters when the build has them.
Interleave builds in a shell loop for A/B claims (box-to-box spread is +-3 %).
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pyorc_amd import _lib, window  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--window", type=int, default=32)
ap.add_argument("--overlap", type=int, default=16)
ap.add_argument("--pairs", type=int, default=1000)
ap.add_argument("--dtype", default="u8")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--warm", type=int, default=6, help="untimed launches first: the first ~5 after an idle gap run slow (tools/clock_ramp.py)")
ap.add_argument("--seed", type=int, default=20260927 + 2)
ap.add_argument("--tag", default="")
a = ap.parse_args()
lib = _lib.load()
_lib.require_device()
T, H, W = a.pairs + 1, a.height, a.width
nr, nc = window.get_array_shape((H, W), (a.window, a.window), (a.overlap, a.overlap))
d_f, d_o = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * H * W))
_lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * a.pairs * nr * nc))
_lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, a.seed, 0.02))
_lib.check(lib.lspiv_synchronize())
code, d_in = 0, d_f
if a.dtype == "f32":   # what the projection hands over: float32 samples (here: the uint8 values)
    from pyorc_amd.device import DeviceFrames
    host = np.empty((T, H, W), np.uint8)
    _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(host), d_f, host.nbytes))
    _lib.check(lib.lspiv_dev_free(d_f))
    dev = DeviceFrames.from_host(host.astype(np.float32))
    code, d_in = 1, dev.c_ptr


def go():
    _lib.check(lib.lspiv_piv_pairs_dev(d_in, code, T, H, W, a.window, a.window, a.overlap, a.overlap, -1.0, d_o, None, None))


for _ in range(max(1, a.warm)):
    go()
_lib.check(lib.lspiv_synchronize())
ev0, ev1 = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_event_create(C.byref(ev0)))
_lib.check(lib.lspiv_event_create(C.byref(ev1)))
_lib.check(lib.lspiv_event_record(ev0))
for _ in range(a.reps):
    go()
_lib.check(lib.lspiv_event_record(ev1))
_lib.check(lib.lspiv_synchronize())
ms = C.c_float()
_lib.check(lib.lspiv_event_elapsed_ms(ev0, ev1, C.byref(ms)))
ms = ms.value / a.reps
extra = ""
if hasattr(lib, "lspiv_rescue_stats") and lib.lspiv_rescue_stats.argtypes:
    st = (C.c_int64 * 5)()
    _lib.check(lib.lspiv_rescue_stats(None, st))
    extra = f" rescue fit {st[0]} amb {st[1]} of {a.pairs * nr * nc}"
print(f"{a.tag or os.environ.get('LSPIV_LIBRARY', 'default')}: {a.window}/{a.overlap} {a.dtype} P={a.pairs}: {ms:.3f} ms -> "
      f"{a.pairs / ms * 1e3:.0f} pairs/s{extra}", flush=True)
