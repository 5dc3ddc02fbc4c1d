#!/usr/bin/env python3
"""Development probe (GPU box): render the small scenes on the GPU and compare with the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import raytracer_amd as R
from raytracer_amd import runtime, compare_images as CI
from oracle import rt_oracle as O

r = runtime.Renderer(0)
cases = [("pool", 96, 96), ("cubes", 96, 54), ("transparency_v7", 160, 90), ("transparency_original", 96, 96)]
if len(sys.argv) > 1:
    cases = [(a.split(":")[0], int(a.split(":")[1]), int(a.split(":")[2])) for a in sys.argv[1:]]
for name, W, H in cases:
    sc = R.parse_scene_file(f"scenes/{name}.txt")
    r.set_scene(sc)
    f32, u8, st = r.render_rows(W, H, seed=0)
    t0 = time.time(); ref, cnt = O.render_frame(sc, W, H, seed=0, threads=16); t1 = time.time()
    m = CI.metrics(np.uint8(ref), u8)
    ok_cnt = (st["primary_rays"], st["secondary_rays"], st["depth_cut"], st["shaded_points"], st["shadow_rays"]) == \
             (cnt["primary"], cnt["secondary"], cnt["depth_cut"], cnt["shaded"], cnt["shadow"])
    print(f"{name} {W}x{H}: within1={100*m['within_1lsb']:.4f}% beyond={m['pixels_beyond_1lsb']} max={m['max_abs_diff']} "
          f"psnr={m['psnr']:.2f} counters_equal={ok_cnt} gpu_ms={st['ms_total']:.3f} (trace {st['ms_trace']:.3f} shade {st['ms_shade']:.3f}) "
          f"oracle_s={t1-t0:.2f} maxabs_f32={np.abs(ref - f32).max():.4f}", flush=True)
    # chain: every nearest-hit query bit for bit
    hb = r.debug_hits(W, H)
    _, _, oh = O.render_pixels(sc, W, H, np.tile(np.arange(W), H), np.repeat(np.arange(H), W), seed=0, record_hits=True)
    key_o = {(h[0], h[1]): h for h in oh}
    bad = 0
    for i in range(len(hb["pixel"])):
        o = key_o.get((int(hb["pixel"][i]), int(hb["node"][i])))
        if o is None or o[2] != hb["surf"][i] or o[3] != hb["t"][i] or tuple(hb["P"][i]) != o[4] or tuple(hb["n"][i]) != o[5]:
            bad += 1
    print(f"   chain: {hb['count']} queries on GPU, {len(oh)} in oracle, mismatching records: {bad}", flush=True)
