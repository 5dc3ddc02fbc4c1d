#!/usr/bin/env python3
"""Development probe (GPU box): which pairs of torch streams run the frames of two handles concurrently."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as R
from raytracer_amd import runtime

W, H = 3840, 2160
scene = R.parse_scene_file("scenes/pool_n16.txt")
first = runtime.Renderer(0); first.set_scene(scene)
streams = [torch.cuda.Stream() for _ in range(8)]
second = runtime.Renderer(0); second.set_scene(scene)
rs = [first, second]
tiles = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
K = 12
for pair in [(4, 5), (0, 1), (5, 6), (6, 7), (7, 0), (0, 1), (1, 3), (4, 5)]:
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            i = k % 2
            rs[i].enqueue_rows_device(W, H, 0, tiles[i].data_ptr(), stream=streams[pair[i]].cuda_stream, seed=k)
        sts = [r.wait() for r in rs]
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3 / K)
    print(f"streams {pair}: {best:.3f} ms per frame, shade events {sum(s['ms_shade'] for s in sts) / K:.3f}", flush=True)
