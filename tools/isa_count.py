#!/usr/bin/env python
"""Instruction mix of one kernel of a translation unit: python tools/isa_count.py <file.hip> <mangled-substring> [-D...]"""
import collections
import re
import subprocess
import sys

src, key = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
asm = "/tmp/isa_count.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                src, "-o", asm] + extra, check=True, capture_output=True)
txt = open(asm).read()
# split into functions
funcs = re.split(r"\n(?=_Z[\w]+:[^\n]*\n)", txt)
for f in funcs:
    name = f.split(":", 1)[0]
    if key not in name or not name.startswith("_Z"):
        continue
    body = f.split("s_endpgm")[0]
    c = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        if not line or line.startswith((";", ".")) or line.endswith(":"):
            continue
        op = line.split()[0]
        if op.startswith("v_"):
            c["valu"] += 1
            if "dpp" in line:
                c["dpp"] += 1
        elif op.startswith("ds_"):
            c["ds"] += 1
        elif op.startswith("scratch_"):
            c["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        c["total"] += 1
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(dem[:100], dict(c))
