#!/usr/bin/env python3
"""Development probe (GPU box): unusual parameters (N not a power of two, N >= 256, many lights, deep mirrors) vs the oracle."""
import os, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as R
from raytracer_amd import runtime, compare_images as CI
from oracle import rt_oracle as O

BASE = ("cam 0 1.2 5  0 0.8 0  0 1 0  1.4 1.6\nset 0.2 0.3 0.5 {N} {R}\n"
        "mtl 0.6 0.7 0.8  1 1 1  0.9 0.9 0.9  8 0.5\nmtl 0.7 0.6 0.4  0.5 0.5 0.5  0.8 0.8 0.8  30 0\n"
        "pln 0 1 0 -0.5 2\npln 0 0 1 -4 2\nsph 0.4 1.0 0.5 1.1 1\nsph -1.8 0.7 -1 0.6 2\nbox 1.5 0.2 1.5 1.0 2\n{LIGHTS}")
def lights(n):
    rng = np.random.default_rng(5)
    return "".join(f"lgt {rng.uniform(-4,4):.3f} {rng.uniform(2,5):.3f} {rng.uniform(-2,5):.3f}  1 0.9 0.8  1 {rng.uniform(0.3,1):.2f} {rng.uniform(0.3,2):.2f}\n" for _ in range(n))
r = runtime.Renderer(0)
for N, Rr, L, w, h in [(17, 4, 2, 64, 40), (7, 31, 1, 48, 32), (300, 2, 1, 16, 10), (3, 3, 12, 64, 40), (255, 2, 1, 16, 10), (1, 10, 3, 96, 64)]:
    text = BASE.format(N=N, R=Rr, LIGHTS=lights(L))
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(text)
    sc = R.parse_scene_file(f.name); os.unlink(f.name)
    r.set_scene(sc)
    ref, cnt = O.render_frame(sc, w, h, seed=2, threads=16)
    for flags in (0, 1):
        _, u8, st = r.render_rows(w, h, seed=2, flags=flags)
        m = CI.metrics(np.uint8(ref), u8)
        ok = (st["primary_rays"], st["secondary_rays"], st["depth_cut"], st["shaded_points"], st["shadow_rays"]) == \
             (cnt["primary"], cnt["secondary"], cnt["depth_cut"], cnt["shaded"], cnt["shadow"])
        print(f"N={N} R={Rr} L={L} flags={flags}: beyond={m['pixels_beyond_1lsb']} max={m['max_abs_diff']} counters_equal={ok} pairs={st['wave_light'][7]}", flush=True)
