#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>_<workload>/ (written by tools/profile.sh on the GPU box) into profiles/.

    python tools/summarize_profile.py r02 [workload]

Writes profiles/<tag>_<workload>_kernel_stats.csv (+ _isolated: rocprofv3 --kernel-trace --stats),
profiles/<tag>_<workload>_pmc.json (per-kernel averages of every counter collected) and updates
profiles/shade_roofline.json, which bench.py reads: per workload, the dominant kernel's executed FP32 flop per
launch, HBM bytes per launch and VALU issue figures, with the hash of the kernel sources they were collected on.

Executed FP32 flop = 64 lanes x SQ_INSTS_VALU_FLOPS_FP32.  That counter adds, per wave-level instruction, 1 for
v_add / v_mul / transcendental, 2 for v_fma / v_pk_add / v_pk_mul and 4 for v_pk_fma_f32 (calibrated on instruction
streams of known length: tools/pmc_calibrate.sh, profiles/r02_pmc_calibration.txt); compares, min / max, conversions and
integer work count nothing.  Every lane is counted (an upper bound of the useful flop).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size, so reads
are doubled before being compared with byte counts.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
wl = sys.argv[2] if len(sys.argv) > 2 else "pool_n16_4k"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{wl}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def newest(pattern):
    """gpurun merges every run's files into the same directories (rocprofv3 names them by process id): only the latest counts"""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


for sub, suffix in (("trace", ""), ("trace_isolated", "_isolated")):
    stats = newest(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_{wl}_kernel_stats{suffix}.csv"))

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
dur64 = collections.defaultdict(list)
for name in ("pmc_sq", "pmc_class", "pmc_fetch", "pmc_write", "pmc_f64"):
    for f in newest(os.path.join(src, name, "*", "*_counter_collection.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if name in ("pmc_fetch", "pmc_f64") and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                (dur if name == "pmc_fetch" else dur64)[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {}
for k, cs in pmc.items():
    if not k.startswith("k_"):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["launches_per_pass"] = len(next(iter(cs.values())))
    if dur[k]:
        out[k]["ms_under_pmc"] = sum(dur[k]) / len(dur[k])
json.dump(out, open(os.path.join(dst, f"{tag}_{wl}_pmc.json"), "w"), indent=1, sort_keys=True)

# the float64 ray-tree kernel: float64 flop executed per launch (add / mul 1, fma 2 per lane; the transcendental seeds of
# the division and square-root expansions count 1) against the vector float64 rate of the chip
trace = {k: c for k, c in out.items() if k.startswith("k_trace") and "SQ_INSTS_VALU_FMA_F64" in c}
if trace:
    kname = max(trace, key=lambda k: trace[k].get("SQ_INSTS_VALU", 0.0))
    c = out[kname]
    g = lambda n: c.get(n, 0.0)
    ms64 = sum(dur64[kname]) / len(dur64[kname]) if dur64[kname] else 0.0
    flop64 = 64.0 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + 2.0 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64"))
    peak64 = 78.6   # TFLOP/s: 1024 SIMDs x 64 lanes x 2 flop per v_fma_f64 every 4 cycles at 2.4 GHz (measured: 3.9 cycles, tools/microbench/f64_cycles.hip)
    entry64 = {
        "kernel": kname, "profile": f"{tag}_{wl}", "kernel_source_hash": kernel_source_hash(),
        "executed_fp64_flop_per_launch": flop64, "ms_under_pmc": ms64,
        "achieved_tflops": flop64 / (ms64 * 1e-3) / 1e12 if ms64 else None, "peak_tflops": peak64,
        "frac": flop64 / (ms64 * 1e-3) / 1e12 / peak64 if ms64 else None,
        "fp64_share_of_valu_insts": (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64")) / g("SQ_INSTS_VALU") if g("SQ_INSTS_VALU") else None,
        "counters": {n: g(n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
                                       "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES") if n in c},
        "note": "latency-bound by construction: a ray's levels depend on each other, every level ends in an IEEE division "
                "(~41 cycles per SIMD) or square root (~69) -- see tools/microbench/f64_cycles.hip",
    }
    path = os.path.join(dst, "trace_roofline.json")
    allw = json.load(open(path)) if os.path.exists(path) else {}
    allw[wl] = entry64
    json.dump(allw, open(path, "w"), indent=1, sort_keys=True)

# the dominant kernel: the shading kernel with the most VALU instructions
shade = {k: c for k, c in out.items() if k.startswith(("k_shade_fast", "k_shade_general"))}
if shade:
    kname = max(shade, key=lambda k: shade[k].get("SQ_INSTS_VALU", 0.0))
    c = out[kname]
    g = lambda n: c.get(n, 0.0)
    fetch, write, ms = g("FETCH_SIZE"), g("WRITE_SIZE"), g("ms_under_pmc")
    cyc = g("GRBM_GUI_ACTIVE") / 8.0          # the counter is summed over the 8 XCDs
    valu = g("SQ_INSTS_VALU")
    flop = 64.0 * g("SQ_INSTS_VALU_FLOPS_FP32")
    entry = {
        "kernel": kname, "profile": f"{tag}_{wl}", "kernel_source_hash": kernel_source_hash(),
        "executed_fp32_flop_per_launch": flop,
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "ms_under_pmc": ms,
        "counters": {n: g(n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
                                       "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32",
                                       "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_FLOPS_FP32", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU",
                                       "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE") if n in c},
        "valu_issue": {"cycles_per_valu_inst_per_simd": (cyc * 1024.0 / valu) if valu else None,
                       "effective_clock_ghz": (cyc / (ms * 1e-3) / 1e9) if ms else None,
                       "fp32_share_of_valu_insts": (g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_TRANS_F32") +
                                                    g("SQ_INSTS_VALU_FMA_F32")) / valu if valu else None,
                       "note": "in-kernel measurement of this chip (profiles/r02_valu_cycles.txt): v_fma_f32 3.7-4.0, v_pk_fma_f32 4.2, "
                               "v_add/mul_f32 2.9, v_cvt / v_mul_lo_u32 / v_max_f32 4.2, v_xor / v_mov 2.3 cycles per wave instruction per SIMD"},
        "note": "FETCH_SIZE doubled (gfx950 counts wide reads at half size); WRITE_SIZE as reported",
    }
    path = os.path.join(dst, "shade_roofline.json")
    allw = json.load(open(path)) if os.path.exists(path) else {}
    allw[wl] = entry
    json.dump(allw, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry, indent=1))
