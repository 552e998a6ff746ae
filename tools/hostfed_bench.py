"""Host-fed throughput of lspiv_piv_pairs (frames in pageable host memory -> results in host memory).
This is the PCIe-inclusive rate of DESIGN.md section 4; it is never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyorc_amd

T = int(sys.argv[1]) if len(sys.argv) > 1 else 301
rng = np.random.default_rng(0)
base = (rng.random((1080, 1920)) ** 6 * 255).astype(np.uint8)
fr = np.stack([np.roll(base, (2 * t, -5 * t), (0, 1)) for t in range(T)])
for dtype in (np.uint8, np.float32, np.float64):
    a = fr if dtype == np.uint8 else fr[: T // (2 if dtype == np.float32 else 4)].astype(dtype)
    pyorc_amd.piv_pairs(a[:3])  # warm-up: context, workspaces
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); pyorc_amd.piv_pairs(a); best = min(best, time.perf_counter() - t0)
    P = a.shape[0] - 1
    print(f"{np.dtype(dtype).name}: {P} pairs in {best*1e3:.1f} ms -> {P/best:.0f} pairs/s, {a.nbytes/best/1e9:.1f} GB/s of host frames")
pin = pyorc_amd.pinned_empty(fr.shape, np.uint8); pin[...] = fr
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); pyorc_amd.piv_pairs(pin); best = min(best, time.perf_counter() - t0)
print(f"uint8, stack in pinned host memory: {T-1} pairs in {best*1e3:.1f} ms -> {(T-1)/best:.0f} pairs/s, {pin.nbytes/best/1e9:.1f} GB/s")
