#!/usr/bin/env python
"""Instruction mix of a walking kernel PER PHASE of its iteration: the phases are delimited by the s_setprio instructions the kernel
carries (piv_fft_impl.h, LSPIV_SETPRIO: every phase outside the four register FFTs runs at priority 1), so the compiler's own
output says where each instruction belongs.  Compile-only.

    python tools/isa_phases.py pyorc_amd/csrc/piv_fft32.hip piv_fft_walk_kernelIhLi32ELb0ELb0 [-D...]

Prints, for the loop body (from the loop header label to the backward branch), one row per segment between two s_setprio: VALU (and how
many of them are v_fma / v_mul / v_add / v_sub = arithmetic of the transforms, v_cvt, v_mov / v_accvgpr, DPP, v_cndmask, v_perm*,
transcendental), LDS, VMEM, scratch, SALU, waitcnt."""
import collections
import re
import subprocess
import sys

src, key = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
asm = "/tmp/isa_phases.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                src, "-o", asm] + extra, check=True, capture_output=True)
txt = open(asm).read()
m = re.search(r"\n(_Z\w*" + re.escape(key) + r"\w*):[^\n]*\n(.*?)s_endpgm", txt, re.S)
if not m:
    sys.exit(f"no kernel matching {key}")
lines = [l.strip() for l in m.group(2).splitlines()]
lines = [l.split(";")[0].strip() for l in lines]
lines = [l for l in lines if l and (l.endswith(":") or not l.startswith("."))]
# the main loop: the backward branch that spans the most instructions
labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
best = (0, 0, 0)
for i, l in enumerate(lines):
    mm = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", l)
    if mm:
        tgt = mm.group(1) or mm.group(2)
        if tgt in labels and labels[tgt] < i and i - labels[tgt] > best[0]:
            best = (i - labels[tgt], labels[tgt], i)
_, lo, hi = best
body = [l for l in lines[lo:hi + 1] if not l.endswith(":")]


def classify(l):
    op = l.split()[0]
    c = collections.Counter()
    if op.startswith("v_"):
        c["valu"] += 1
        if "dpp" in l or "row_" in l or "quad_perm" in l:
            c["dpp"] += 1
        base = op.replace("_e32", "").replace("_e64", "")
        if re.match(r"v_(fma|fmac|mul|add|sub|mad|pk_)", base) and ("f32" in base):
            c["arith_f32"] += 1
        elif base.startswith("v_cvt"):
            c["cvt"] += 1
        elif base.startswith(("v_mov", "v_accvgpr")):
            c["mov"] += 1
        elif base.startswith("v_cndmask"):
            c["cndmask"] += 1
        elif base.startswith(("v_perm", "v_readlane", "v_readfirstlane", "v_writelane", "v_bfe", "v_lshl", "v_lshr", "v_and", "v_or", "v_xor", "v_bfi", "v_alignbit", "v_add_u32", "v_sub_u32", "v_add_co", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_ashr", "v_dot4")):
            c["int/bit"] += 1
        elif base.startswith(("v_max", "v_min", "v_med3", "v_cmp")):
            c["minmax/cmp"] += 1
        elif base.startswith(("v_log", "v_exp", "v_rcp", "v_rsq", "v_sqrt")):
            c["trans"] += 1
        else:
            c["other_valu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
        if "bpermute" in op or "swizzle" in op:
            c["lds_perm"] += 1
    elif op.startswith("scratch_"):
        c["scratch"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")):
        c["vmem"] += 1
    elif op.startswith("s_waitcnt"):
        c["waitcnt"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    return c


segs, cur, prio = [], collections.Counter(), "?"
for l in body:
    if l.startswith("s_setprio"):
        segs.append((prio, cur))
        cur, prio = collections.Counter(), l.split()[1]
        continue
    cur += classify(l)
segs.append((prio, cur))
cols = ["valu", "arith_f32", "cvt", "mov", "cndmask", "int/bit", "minmax/cmp", "trans", "dpp", "other_valu", "lds", "lds_perm", "vmem", "scratch", "salu", "waitcnt"]
print(f"loop body: {len(body)} instructions, {len(segs)} segments (priority after the s_setprio that opens the segment)")
print("seg prio " + " ".join(f"{c:>10s}" for c in cols))
tot = collections.Counter()
for k, (p, c) in enumerate(segs):
    tot += c
    print(f"{k:3d} {p:>4s} " + " ".join(f"{c[x]:10d}" for x in cols))
print("    all " + " ".join(f"{tot[x]:10d}" for x in cols))
