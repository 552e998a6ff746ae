#!/usr/bin/env python
"""Per-kernel register / scratch / occupancy table of one kernel translation unit (hipcc -Rpass-analysis).

    python tools/kres.py pyorc_amd/csrc/piv_fft32.hip [filter-substring] [-D...]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-c", src, "-o", "/tmp/kres.o",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("sgpr", r"SGPRs: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("lspiv::", "").replace("unsigned char", "u8").replace("(PivParams)", "").replace("void ", "")
    if flt and not all(f in n for f in flt):
        continue
    print(f"{n:70s} vgpr {r.get('vgpr'):4d} agpr {r.get('agpr', 0):3d} scratch {r.get('scratch'):4d} occ {r.get('occ')}")
if "error" in out:
    print(out[-3000:])
