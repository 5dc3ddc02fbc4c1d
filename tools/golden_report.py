#!/usr/bin/env python3
"""Development probe (GPU box): parity figures against the committed reference goldens (frames and sparse pixels)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import golden_npz, golden_png, scene
from raytracer_amd import runtime, compare_images as CI
r = runtime.Renderer(0)
for name, w, h in [("pool", 500, 500), ("cubes", 480, 270), ("transparency_v7", 480, 270), ("transparency_original", 256, 256)]:
    r.set_scene(scene(name))
    _, u8, _ = r.render_rows(w, h, seed=0)
    m = CI.metrics(golden_png(f"frames/{name}_{w}x{h}_seed0.png"), u8)
    print(f"frame {name} {w}x{h}: {100 * m['within_1lsb']:.4f} % within 1 LSB, {m['pixels_beyond_1lsb']} beyond, max {m['max_abs_diff']}", flush=True)
for key in ["cubes_1920x1080", "transparency_v7_1920x1080", "pool_3840x2160", "pool_n16_3840x2160", "synthetic_10k_1k_3840x2160"]:
    z = golden_npz(f"sparse/{key}.npz")
    r.set_scene(scene(key.rsplit("_", 1)[0]))
    _, u8, _ = r.render_rows(int(z["width"]), int(z["height"]), seed=int(z["seed"]), want_f32=False)
    m = CI.metrics(np.uint8(z["rgb"]), u8[z["y"], z["x"]])
    print(f"sparse {key}: {len(z['x'])} pixels, {m['pixels_beyond_1lsb']} beyond 1 LSB, max {m['max_abs_diff']}", flush=True)
