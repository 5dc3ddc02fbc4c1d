#!/usr/bin/env python3
"""Development probe (GPU box, library built with -DRT_EXP_SPAN): when do the waves of k_shade_fast start and finish?
    RT_AMD_LIB=.../librt_span.so python tools/span_probe.py [world ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as R
from raytracer_amd import runtime
from raytracer_amd.distributed import stripe_rows_for, max_rows_per_rank

W, H = 3840, 2160
scene = R.parse_scene_file("scenes/pool_n16.txt")
r = runtime.Renderer(0)
r.set_scene(scene)
s = torch.cuda.Stream()
for world in [int(a) for a in sys.argv[1:]] or [1, 8]:
    sr = stripe_rows_for(H, world)
    tile = torch.zeros((max_rows_per_rank(H, world, sr), W, 3), dtype=torch.uint8, device="cuda")
    for rep in range(3):
        r.enqueue_rows_device(W, H, 0, tile.data_ptr(), stream=s.cuda_stream, seed=rep, stripe_rows=sr, stripe_count=world, stripe_index=0)
        st = r.wait()
    c = st["wave_light"]
    first = (1 << 44) - c[0]
    n = c[4]
    print(f"world {world}: waves {n}  span {(c[1] - first) / 100:.1f} us  mean start {(c[5] / n - first) / 100:.1f} us  "
          f"mean finish {(c[3] / n - first) / 100:.1f} us  mean resident {c[2] / n / 100:.1f} us  longest item {c[6] / 100:.1f} us  (kernel event: main {st['ms_shade_main'] * 1e3:.1f} us)")
