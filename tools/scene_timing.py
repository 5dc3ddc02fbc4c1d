#!/usr/bin/env python3
"""Development probe (GPU box): frame time of any scene file:  scene_timing.py scene.txt W H"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as R
from raytracer_amd import runtime

path, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
r = runtime.Renderer(0)
r.set_scene(R.parse_scene_file(path))
out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
best = None
for it in range(6):
    st = r.render_rows_device(W, H, 0, out.data_ptr(), stream=stream, seed=it)
    if it and (best is None or st["ms_total"] < best["ms_total"]):
        best = st
print("wave-light classes", best["wave_light"], "bands", best["bands"])
print(f"{os.path.basename(path)} {W}x{H}: total {best['ms_total']:.3f} ms  trace {best['ms_trace']:.3f} shade {best['ms_shade']:.3f}  rays {best['total_rays']}", flush=True)
