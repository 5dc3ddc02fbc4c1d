#!/usr/bin/env python3
"""Development probe: time one rank's share of the headline frame for world sizes 1..8 (single GPU)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as R
from raytracer_amd import runtime
from raytracer_amd.distributed import stripe_rows_for, max_rows_per_rank

W, H = 3840, 2160
r = runtime.Renderer(0)
r.set_scene(R.parse_scene_file("scenes/pool_n16.txt"))
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
for world in worlds:
    sr = stripe_rows_for(H, world)
    tile = torch.zeros((max_rows_per_rank(H, world, sr), W, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for rank in sorted({0, world - 1}):
        ts = []
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st = r.render_rows_device(W, H, 0, tile.data_ptr(), stream=stream, seed=it, stripe_rows=sr, stripe_count=world, stripe_index=rank)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"world {world} rank {rank}: wall {min(ts[1:]):.3f} ms  stream {st['ms_total']:.3f}  trace {st['ms_trace']:.3f} shade {st['ms_shade']:.3f} resolve {st['ms_resolve']:.3f} rays {st['total_rays']}", flush=True)
