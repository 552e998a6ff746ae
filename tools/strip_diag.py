"""Which outputs of an ensemble run depend on LSPIV_STRIP_W (diagnostic for tests/test_gpu_strip_order.py)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:] or ["64", "48", "uint8", "300", "1200", "30", "1"]
tmp = tempfile.mkdtemp()
res = {}
for sw in (None, 0, 7):
    env = dict(os.environ); env.pop("LSPIV_STRIP_W", None)
    if sw is not None: env["LSPIV_STRIP_W"] = str(sw)
    for k, v in [a.split("=") for a in os.environ.get("DIAG_ENV", "").split(",") if a]: env[k] = v
    out = os.path.join(tmp, f"{sw}.npz")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "strip_order_worker.py"), out] + args, check=True, env=env, cwd=ROOT)
    res[sw] = np.load(out)
for sw in (0, 7):
    for k in res[None].files:
        a, b = res[None][k], res[sw][k]
        ne = a.view(np.uint32) != b.view(np.uint32)
        print(args, "strip", sw, k, a.shape, "differing", int(ne.sum()), "max abs", float(np.nanmax(np.abs(a - b))) if ne.any() else 0.0,
              "where", np.argwhere(ne)[:6].tolist())
