#!/usr/bin/env python3
"""Development probe (GPU box): isolate which light / depth of a random scene differs from the oracle."""
import os, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as R
from raytracer_amd import runtime
from oracle import rt_oracle as O
from test_gpu_parity import _random_scene_text

def load(text):
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(text)
    sc = R.parse_scene_file(f.name); os.unlink(f.name)
    return sc

r = runtime.Renderer(0)
seed = int(sys.argv[1])
text = _random_scene_text(5000 + seed, int(os.environ.get('FAMILY', '0')))
lines = text.strip().split("\n")
lights = [i for i, l in enumerate(lines) if l.startswith("lgt")]
setl = [i for i, l in enumerate(lines) if l.startswith("set")][0]
w, h = 96, 64
def run(tag, ls):
    sc = load("\n".join(ls) + "\n")
    r.set_scene(sc)
    _, u8, _ = r.render_rows(w, h, seed=seed)
    ref, _ = O.render_frame(sc, w, h, seed=seed, threads=16)
    d = np.abs(np.uint8(ref).astype(int) - u8.astype(int)).max(axis=2)
    ys, xs = np.nonzero(d > 1)
    print(f"{tag}: beyond={len(ys)} max={d.max()}", [(int(x), int(y)) for x, y in zip(xs[:4], ys[:4])], flush=True)
    return ref, u8
run("full", lines)
for depth in (1, 2):
    ls = list(lines); p = ls[setl].split(); p[5] = str(depth); ls[setl] = " ".join(p)
    run(f"depth {depth}", ls)
for k in lights:
    ls = [l for i, l in enumerate(lines) if i not in lights or i == k]
    run(f"only light {k - lights[0]}", ls)
    ls1 = list(ls); p = ls1[setl].split(); p[5] = "1"; ls1[setl] = " ".join(p)
    run(f"only light {k - lights[0]} depth 1", ls1)
