#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into profiles/.

    python tools/summarize_profile.py r01

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), profiles/<tag>_pmc.json (per-kernel
averages of every counter collected) and profiles/shade_hbm_traffic.json, which bench.py reads for
roofline.traffic.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half
their size, so reads are doubled before being compared with byte counts.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
stats = glob.glob(os.path.join(src, "trace_isolated", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats_isolated.csv"))

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for name in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if name == "pmc_fetch" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {}
for k, cs in pmc.items():
    if not k.startswith("k_"):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["launches_per_pass"] = len(next(iter(cs.values())))
    if dur[k]:
        out[k]["ms_under_pmc"] = sum(dur[k]) / len(dur[k])
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)

kname = next((k for k in out if k.startswith("k_shade_fast")), None)
if kname:
    c = out[kname]
    fetch, write = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
    ms = c.get("ms_under_pmc", 0.0)
    clk = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (ms * 1e-3) / 1e9 if ms else None  # counter is summed over 8 XCDs
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    valu = c.get("SQ_INSTS_VALU", 0.0)
    summary = {
        "kernel": kname, "profile": tag,
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "fetch_size_kib": fetch, "write_size_kib": write,
        "note": "FETCH_SIZE doubled (gfx950 counts wide reads at half size); WRITE_SIZE as reported",
        "valu_insts_per_launch": valu, "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
        "ms_under_pmc": ms, "effective_clock_ghz": clk,
        "cycles_per_valu_inst_per_simd": (cyc * 1024.0 / valu) if valu else None,
    }
    json.dump(summary, open(os.path.join(dst, "shade_hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))
