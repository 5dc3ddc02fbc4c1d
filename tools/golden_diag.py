#!/usr/bin/env python3
"""Development probe (GPU box): the pixels of a golden frame that differ by more than 1 LSB, with the oracle's hit records."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import golden_png, scene
from raytracer_amd import runtime
from oracle import rt_oracle as O
r = runtime.Renderer(0)
for name, w, h in [("pool", 500, 500), ("cubes", 480, 270)]:
    sc = scene(name)
    r.set_scene(sc)
    gold = golden_png(f"frames/{name}_{w}x{h}_seed0.png")
    for flags in (0, 1, 2):
        _, u8, _ = r.render_rows(w, h, seed=0, flags=flags)
        d = np.abs(gold.astype(int) - u8.astype(int)).max(axis=2)
        ys, xs = np.nonzero(d > 1)
        print(name, "flags", flags, "beyond", len(ys), [(int(x), int(y)) for x, y in zip(xs, ys)])
        if flags == 0 and len(ys):
            c, _, hits = O.render_pixels(sc, w, h, xs.astype(np.int32), ys.astype(np.int32), seed=0, record_hits=True)
            for x, y in zip(xs, ys):
                print("  px", (int(x), int(y)), "gold", gold[y, x], "gpu", u8[y, x])
            for o in hits:
                print("   hit pixel", o[0], "node", o[1], "surf", o[2], "t %.4f" % o[3], "P", tuple(round(v, 4) for v in o[4]), "n", tuple(round(v, 3) for v in o[5]))
