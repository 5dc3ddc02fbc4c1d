#!/usr/bin/env python3
"""Times the UNMODIFIED reference renderer on this machine's CPU cores (build container only).

TEST INFRASTRUCTURE.  Usage:  python -m oracle.time_reference [--seconds 45] [--procs 8]

For every benchmark configuration's scene, with that configuration's own N and max_recursions, at a reduced
resolution of the same aspect ratio: (a) ONE process renders interleaved rows for about --seconds, (b) --procs
processes render disjoint interleaved row sets of the same frame at the same time.  The reference's own loop body
is what runs -- camera ray, trace_ray, clip (/root/reference/ray_tracer.py:319-331) -- with its own Mersenne-Twister
jitter; only counting wrappers sit around it (oracle/ref_harness.py), so the figure includes a few Python calls per
ray-tree node that the reference itself does not make.  Rays are counted as everywhere in this repository: primary +
traced secondary + N*N (or 1) shadow rays per (shaded hit, light).  Writes tests/golden/ref_cpu_baseline.json: data
only (seconds, counters, rays per second); nothing of the reference is copied.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (workload of bench.py, scene, full resolution, reduced resolution used here)
CONFIGS = [
    ("pool_500 (BASELINE config #1)", "pool", (500, 500), (96, 96)),
    ("cubes_1080p (config #2)", "cubes", (1920, 1080), (160, 90)),
    ("v7_1080p (config #3)", "transparency_v7", (1920, 1080), (160, 90)),
    ("synthetic_4k (config #4)", "synthetic_10k_1k", (3840, 2160), (32, 18)),
    ("pool_n16_4k (config #5, headline)", "pool_n16", (3840, 2160), (96, 54)),
]


def _render_rows(args):
    """Rows y = first, first + step, ... of a (w, h) frame, for at most `budget` seconds; returns (seconds, rows done, counters)."""
    name, w, h, first, step, budget = args
    from oracle import ref_harness as H
    rt = H.load()["rt"]
    scene = H.parse(os.path.join(ROOT, "scenes", name + ".txt"))
    n_shadow = int(scene.root_number_shadow_rays)
    np.random.seed(1234 + first)
    done = 0
    with H.instrumented(scene, 0, counter_jitter=False) as probe:
        t0 = time.perf_counter()
        for y in range(first, h, step):
            for x in range(w):
                o, d = scene.camera.get_ray(x, y, w, h)
                c = rt.trace_ray(o, d, scene, n_shadow, current_depth=0)
                np.clip(c, 0, 1) * 255
            done += 1
            if time.perf_counter() - t0 > budget:
                break
        dt = time.perf_counter() - t0
        counters = {"primary": probe.primary, "secondary": probe.secondary, "shaded": probe.shaded, "shadow": probe.shadow}
    return dt, done, counters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=45.0)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    out = {"host": {"cpu": platform.processor() or platform.machine(), "logical_cpus": os.cpu_count(), "python": platform.python_version(),
                    "numpy": np.__version__},
           "ray_definition": "primary + traced secondary + N*N (or 1) shadow rays per (shaded hit, light)",
           "note": "unmodified reference driven through counting wrappers, its own Mersenne-Twister jitter; reduced resolution, "
                   "the configuration's own N and max_recursions; each run stops after the row that passes the time budget",
           "runs": {}}
    path = os.path.join(ROOT, "tests", "golden", "ref_cpu_baseline.json")
    if a.only and os.path.exists(path):
        with open(path) as f:
            out["runs"] = json.load(f).get("runs", {})
    for label, name, full, (w, h) in CONFIGS:
        if a.only and not any(k in label for k in a.only):
            continue
        entry = {"scene": name, "full_resolution": list(full), "resolution_timed": [w, h]}
        dt, rows, c = _render_rows((name, w, h, 0, 1, a.seconds))
        rays = c["primary"] + c["secondary"] + c["shadow"]
        entry["one_process"] = {"seconds": round(dt, 2), "rows": rows, "counters": c, "rays": rays, "krays_per_s": round(rays / dt / 1e3, 2),
                                "pixels_per_s": round(c["primary"] / dt, 1)}
        print(label, "1 process:", entry["one_process"], flush=True)
        with mp.Pool(a.procs) as pool:
            t0 = time.perf_counter()
            res = pool.map(_render_rows, [(name, w, h, k, a.procs, a.seconds) for k in range(a.procs)])
            wall = time.perf_counter() - t0
        tot = {k: sum(r[2][k] for r in res) for k in ("primary", "secondary", "shaded", "shadow")}
        rays = tot["primary"] + tot["secondary"] + tot["shadow"]
        busy = max(r[0] for r in res)
        entry["processes"] = {"n": a.procs, "seconds_slowest_worker": round(busy, 2), "seconds_wall_incl_start": round(wall, 2), "counters": tot,
                              "rays": rays, "krays_per_s": round(rays / busy / 1e3, 2), "pixels_per_s": round(tot["primary"] / busy, 1)}
        print(label, a.procs, "processes:", entry["processes"], flush=True)
        full_px = full[0] * full[1]
        entry["full_frame_estimate_s"] = {"one_process": round(full_px / entry["one_process"]["pixels_per_s"], 1),
                                          f"{a.procs}_processes": round(full_px / entry["processes"]["pixels_per_s"], 1)}
        out["runs"][label] = entry
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    print("written", path)


if __name__ == "__main__":
    main()
