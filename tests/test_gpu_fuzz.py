"""Randomised differential test on the GPU: tools/fuzz_parity.py (random shapes, dtypes, windows, overlaps, thresholds,
constant patches, empty frames) must report zero gate violations against the C oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_parity(gpu, seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), str(seed), "40"],
                         capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "40 cases, 0 failures" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_callers(gpu, seed):
    """tools/fuzz_modes.py: get_piv per time step (chunked == one call bit for bit, both vs the oracle's get_ffpiv), ensemble
    mode with random chunk sizes / thresholds, and the plane volume, over random sizes, dtypes and shapes."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_modes.py"), str(seed), "60"],
                         capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "60 cases, 0 failures" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [103, 106, 107, 113])
def test_fuzz_ensemble_every_window(gpu, seed):
    """Ensemble mode only, 80 cases per seed, gate: EVERY window of every case within 1e-4 (round 4: the float64 rescue of the
    final fit).  These four seeds each held a case the float32 fit missed -- an 8 x 8 ensemble of two kept pairs, windows on
    the edge of a constant patch whose exact answer is zero displacement."""
    env = dict(os.environ, FUZZ_MODE="ensemble")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_modes.py"), str(seed), "80"],
                         capture_output=True, text=True, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "80 cases, 0 failures" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("tool,seed,n", [("fuzz_parity.py", 701, 60), ("fuzz_modes.py", 711, 60)])
def test_fuzz_on_wide_window_grids(gpu, tool, seed, n):
    """FUZZ_WIDE=1: grids of 26 ... 70 columns -- wider than the walking kernels' job strips (24 / 32 windows), which the small
    frames of the runs above never are."""
    env = dict(os.environ, FUZZ_WIDE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seed), str(n)], capture_output=True, text=True, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert f"{n} cases, 0 failures" in out.stdout


@pytest.mark.gpu
def test_fuzz_rows(gpu):
    """tools/fuzz_rows.py: filters, both projections, masks and int16 packing against their oracles over random shapes
    (odd widths, single frames, tiny frames), dtypes and parameters."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_rows.py"), "31", "150"],
                         capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "150 cases, 0 failures" in out.stdout



@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["LSPIV_NORM_FRAME_MAJOR=1", "LSPIV_BLUR_BLOCK=1", "LSPIV_PROJECT_ONE_CELL=1", "LSPIV_PROJECT_FPT=4", "LSPIV_PROJECT_GX=0"])
def test_fuzz_rows_under_the_ab_switches(gpu, switch):
    """The kernels kept behind environment switches for A/B timing (frame-major normalize, tile-per-block blur, one-cell projection,
    other frames-per-thread / grid shapes) must still give the oracles' bits: a comparison against a broken alternative is worthless."""
    k, v = switch.split("=")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_rows.py"), "41", "80"], capture_output=True, text=True, cwd=ROOT,
                         env=dict(os.environ, **{k: v}))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "80 cases, 0 failures" in out.stdout
