#!/usr/bin/env python3
"""Development probe (GPU box): frame time of a scene through the automatic kernel choice and through the any-size kernel.
    flags_timing.py scene.txt W H"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as R
from raytracer_amd import runtime
path, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
r = runtime.Renderer(0)
r.set_scene(R.parse_scene_file(path))
out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for flags in (0, 1):
    best = None
    for it in range(5):
        st = r.render_rows_device(W, H, 0, out.data_ptr(), stream=stream, seed=it, flags=flags)
        if it and (best is None or st["ms_total"] < best["ms_total"]):
            best = st
    print(f"flags {flags}: total {best['ms_total']:.3f} ms  trace {best['ms_trace']:.3f} shade {best['ms_shade']:.3f}  classes {best['wave_light']}", flush=True)
