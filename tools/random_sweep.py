#!/usr/bin/env python3
"""Development probe (GPU box): many random small scenes, GPU vs oracle; prints the worst cases."""
import os, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as R
from raytracer_amd import runtime, compare_images as CI
from oracle import rt_oracle as O
from test_gpu_parity import _random_scene_text

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
family = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r = runtime.Renderer(0)
bad = 0
tot_beyond = 0
for seed in range(n):
    text = _random_scene_text(5000 + seed, family)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(text)
    sc = R.parse_scene_file(f.name)
    os.unlink(f.name)
    r.set_scene(sc)
    w, h = 96, 64
    _, u8, st = r.render_rows(w, h, seed=seed)
    ref, cnt = O.render_frame(sc, w, h, seed=seed, threads=16)
    m = CI.metrics(np.uint8(ref), u8)
    ok_cnt = (st["primary_rays"], st["secondary_rays"], st["depth_cut"], st["shaded_points"], st["shadow_rays"]) == \
             (cnt["primary"], cnt["secondary"], cnt["depth_cut"], cnt["shaded"], cnt["shadow"])
    tot_beyond += m["pixels_beyond_1lsb"]
    if m["pixels_beyond_1lsb"] > 0 or not ok_cnt:
        bad += 1
        print(f"seed {seed}: beyond={m['pixels_beyond_1lsb']} max={m['max_abs_diff']} counters_equal={ok_cnt} surfaces={len(sc.surfaces)} N={sc.root_number_shadow_rays}", flush=True)
print(f"{n} scenes, {bad} with any pixel beyond 1 LSB, {tot_beyond} such pixels of {n * 96 * 64}")
