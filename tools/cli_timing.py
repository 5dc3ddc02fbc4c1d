#!/usr/bin/env python3
"""Development probe (GPU box): the drop-in render() of configuration #1 (pool.txt, 500 x 500) end to end."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracer_amd as R
sc = R.parse_scene_file("scenes/pool.txt")
for it in range(3):
    t0 = time.perf_counter()
    img = R.render(sc, 500, 500, verbose=False)
    dt = time.perf_counter() - t0
    print(f"render() call {it}: {dt * 1e3:.2f} ms, array {img.shape} {img.dtype}", flush=True)
