"""The C ABI from C: include/lspiv.h is strict C99 and examples/piv_from_c.c, a consumer with no Python in it, builds against
the shipped liblspiv_hip.so; without a device it stops loudly (status 2, no CPU fallback behind the ABI), on the GPU box it
recovers the displacement it drew."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "pyorc_amd")


def build(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    if not os.path.exists(os.path.join(LIBDIR, "liblspiv_hip.so")):
        pytest.skip("liblspiv_hip.so not built")
    exe = str(tmp_path / "piv_from_c")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "piv_from_c.c"), "-L", LIBDIR, "-llspiv_hip", "-lm", f"-Wl,-rpath,{LIBDIR}", "-o", exe],
                   check=True, capture_output=True, text=True)
    return exe


def test_header_is_strict_c99_and_the_c_example_links(tmp_path):
    exe = build(tmp_path)
    from pyorc_amd import _lib

    n = _lib.device_count()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert "ABI 5" in r.stdout
    if n == 0:
        assert r.returncode == 2 and "no gfx950 device" in r.stderr      # loud, not a CPU answer


@pytest.mark.gpu
def test_c_example_recovers_its_displacement(gpu, tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "over 690 vectors" in r.stdout, r.stdout                      # every window of both pairs has a finite result
