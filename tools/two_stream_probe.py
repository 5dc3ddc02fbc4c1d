#!/usr/bin/env python3
"""Development probe (GPU box): frames alternating over two handles on two streams vs one handle on one stream.

    python tools/two_stream_probe.py [world ...]     # one rank's share of the headline frame per world size
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as R
from raytracer_amd import runtime
from raytracer_amd.distributed import stripe_rows_for, max_rows_per_rank, concurrent_streams

W, H = 3840, 2160
scene = R.parse_scene_file("scenes/pool_n16.txt")
one_lane = "--one-lane" in sys.argv
worlds = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 8]
K = 24
for lanes in ((1,) if one_lane else tuple(int(x) for x in os.environ.get('PROBE_LANES', '1,2,3,4').split(','))):
    rs = [runtime.Renderer(0) for _ in range(lanes)]
    for r in rs:
        r.set_scene(scene)
    streams = concurrent_streams(torch, torch.device('cuda', 0), lanes)
    for world in worlds:
        sr = stripe_rows_for(H, world)
        tiles = [torch.zeros((max_rows_per_rank(H, world, sr), W, 3), dtype=torch.uint8, device="cuda") for _ in range(lanes)]
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(K):
                i = k % lanes
                rs[i].enqueue_rows_device(W, H, 0, tiles[i].data_ptr(), stream=streams[i].cuda_stream, seed=k,
                                          stripe_rows=sr, stripe_count=world, stripe_index=0)
            sts = [r.wait() for r in rs]
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3 / K)
        shade = sum(s["ms_shade"] for s in sts) / K
        trace = sum(s["ms_trace"] for s in sts) / K
        print(f"streams {lanes} world {world}: {best:.3f} ms per frame (kernel events: trace {trace:.3f} shade {shade:.3f})", flush=True)
    del rs
