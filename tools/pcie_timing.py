#!/usr/bin/env python3
"""Development probe (GPU box): headline frame through rt_render (uint8 frame copied to the host) vs left on the device."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as R
from raytracer_amd import runtime
W, H = 3840, 2160
r = runtime.Renderer(0)
r.set_scene(R.parse_scene_file("scenes/pool_n16.txt"))
out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
for mode in ("device", "host_u8", "host_f32_u8"):
    ts = []
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "device":
            r.render_rows_device(W, H, 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, seed=it)
        else:
            r.render_rows(W, H, seed=it, want_f32=(mode == "host_f32_u8"))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{mode}: {min(ts[1:]):.2f} ms per frame (wall)", flush=True)

# a sequence of frames through the ring of pinned host buffers: the copy of frame k runs behind the kernels of frame k+1
from raytracer_amd.frames import FrameStream
fs = FrameStream(r, W, H)
n = 12
for it in range(3):
    fs.submit(it)
list(fs.drain()); fs.stats()
torch.cuda.synchronize(); t0 = time.perf_counter()
got = 0
for it in range(n):
    got += fs.submit(100 + it) is not None
got += len(list(fs.drain()))
dt = (time.perf_counter() - t0) * 1e3
fs.stats()
print(f"host_u8_stream: {dt / n:.2f} ms per frame (wall, {got} frames delivered to pinned host memory)", flush=True)
