"""Summarise a tools/profile.sh output directory into profiles/<tag>_summary.json (per-launch means of the dominant kernel).

    python tools/summarize_profile.py gpurun_out/prof_r1d r01_v3

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are collected in
separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of 16-byte-per-lane reads
(which is what the window gather issues), so fetch bytes = 2 * FETCH_SIZE * 1024.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyorc_amd._lib import kernel_code_hash  # noqa: E402  (the same function bench.py checks a summary against)

import re

src, tag = sys.argv[1], sys.argv[2]
# SUMMARY_KERNELS: regex of the kernels to summarise (default: the fused PIV kernels); SUMMARY_ROW_FRAMES / SUMMARY_CALLS: a row of
# tools/rows_launch.py -- frames per call and calls made, so that kernels launched several times per call are weighted
KRE = re.compile(os.environ.get("SUMMARY_KERNELS", "piv_"))
# optional: the launch shape the profile was taken on (bench.py defaults), so bench.py can match it: pairs H W window overlap
shape = [int(x) for x in sys.argv[3:8]] if len(sys.argv) >= 8 else [1000, 1080, 1920, 32, 16]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if KRE.search(r["Kernel_Name"]):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
kernels = sorted({k for k, _ in acc})
stats = {}
for r in csv.DictReader(open(os.path.join(src, "trace_kernel_stats.csv"))):
    if KRE.search(r["Name"]):
        stats[r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"])}
# code_hash: the PIV kernel sources this profile was taken on; bench.py reports `traffic: null` for a summary whose hash is not
# the tree's (a number measured on other kernel code is not this kernel's traffic)
out = {"tag": tag, "source": src, "launch": dict(zip(("pairs", "H", "W", "window", "overlap"), shape)), "code_hash": kernel_code_hash(),
       "kernels": {}}
for k in kernels:
    c = {n: sum(v) / len(v) for (kk, n), v in acc.items() if kk == k}
    d = {"counters_mean_per_launch": c, "trace": stats.get(k)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and os.environ.get("SUMMARY_ROW_FRAMES"):
        # a row of tools/rows_launch.py: FETCH_SIZE tallies 128-byte requests as 64 bytes (the guide's gfx950 note: exactly half for wide
        # coalesced reads) but narrower accesses in full, and which of the two a kernel's loads produce is not documented -- calibrated per
        # kernel on what it cannot avoid reading: a raw figure below 3/4 of the compulsory input bytes is a halved one (x 2), anything
        # else is taken as it is.  WRITE_SIZE needs nothing: time_diff writes exactly its output bytes by it.
        from tools.rows_launch import row_read_bytes  # noqa: E402

        raw = c["FETCH_SIZE"] * 1024.0
        must = float(row_read_bytes(os.environ.get("SUMMARY_ROW", ""), int(os.environ["SUMMARY_ROW_FRAMES"])))
        scale = 2.0 if 0.3 * must < raw < 0.75 * must else 1.0     # (a kernel that reads a small part of the input -- normalize's sampled mean -- is left alone)
        d["fetch_size_raw_bytes"], d["compulsory_read_bytes_of_the_row"], d["fetch_scale"] = raw, must, scale
        d["hbm_fetch_bytes"] = scale * raw
        d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024.0
        d["hbm_traffic_bytes"] = d["hbm_fetch_bytes"] + d["hbm_write_bytes"]
    elif "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_fetch_bytes"] = 2.0 * c["FETCH_SIZE"] * 1024.0
        d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024.0
        d["hbm_traffic_bytes"] = d["hbm_fetch_bytes"] + d["hbm_write_bytes"]
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        d["valu_inst_per_simd_per_4cyc"] = c["SQ_INSTS_VALU"] / (1024.0 * cycles / 4.0)
        d["valu_inst_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    out["kernels"][k] = d
if os.environ.get("SUMMARY_ROW_FRAMES"):
    from tools.rows_launch import rows_source_hash  # noqa: E402

    out["launch"] = {"frames": int(os.environ["SUMMARY_ROW_FRAMES"]), "H": 1080, "W": 1920}
    out["rows_source_hash"] = rows_source_hash()
    calls = float(os.environ.get("SUMMARY_CALLS", "0"))
    for k, d in out["kernels"].items():
        if d.get("trace") and calls:
            d["launches_per_call"] = d["trace"]["calls"] / calls
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_summary.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
