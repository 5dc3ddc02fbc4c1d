#!/usr/bin/env python3
"""Development probe (GPU box): one random scene of tools/random_sweep.py in detail."""
import os, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as R
from raytracer_amd import runtime, compare_images as CI
from oracle import rt_oracle as O
from test_gpu_parity import _random_scene_text

r = runtime.Renderer(0)
family = int(os.environ.get('FAMILY', '0'))
for seed in [int(a) for a in sys.argv[1:]]:
    text = _random_scene_text(5000 + seed, family)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(text)
    sc = R.parse_scene_file(f.name)
    os.unlink(f.name)
    print(f"==== seed {seed}\n{text}")
    r.set_scene(sc)
    w, h = 96, 64
    ref, cnt = O.render_frame(sc, w, h, seed=seed, threads=16)
    for flags in (0, 2, 1):
        _, u8, st = r.render_rows(w, h, seed=seed, flags=flags)
        d = np.abs(np.uint8(ref).astype(int) - u8.astype(int)).max(axis=2)
        ys, xs = np.nonzero(d > 1)
        print(f"flags {flags}: beyond={len(ys)} max={d.max()}")
        if flags == 0:
            bad = list(zip(xs.tolist(), ys.tolist()))
    if bad:
        xs = np.array([b[0] for b in bad]); ys = np.array([b[1] for b in bad])
        _, _, oh = O.render_pixels(sc, w, h, xs, ys, seed=seed, record_hits=True)
        prim = {o[0]: o for o in oh if o[1] == 1}
        from collections import Counter
        print("primary surfaces of bad pixels:", Counter(prim[y * w + x][2] for x, y in bad))
        _, u8, _ = r.render_rows(w, h, seed=seed)
        for x, y in bad[:8]:
            o = prim[y * w + x]
            print(f"  px ({x},{y}) surf {o[2]} t {o[3]:.6f} P {tuple(round(v, 4) for v in o[4])} n {tuple(round(v, 4) for v in o[5])} ref {np.uint8(ref)[y, x]} gpu {u8[y, x]}")
