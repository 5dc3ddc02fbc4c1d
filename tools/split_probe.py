#!/usr/bin/env python3
"""Development probe (GPU box): classification / sampling kernel times of the headline frame, split against fused.

    python tools/split_probe.py [workload]      (RT_SPLIT=0 selects the fused kernel)
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as R
from raytracer_amd import runtime
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "pool_n16_4k"
name, W, H = WORKLOADS[wl]
r = runtime.Renderer(0)
r.set_scene(R.parse_scene_file(f"scenes/{name}.txt"))
u8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
for i in range(5):
    st = r.render_rows_device(W, H, 0, u8.data_ptr(), seed=i)
print(wl, "RT_SPLIT=" + os.environ.get("RT_SPLIT", "1"), {k: round(st[k], 3) for k in ("ms_trace", "ms_shade", "ms_shade_main", "ms_shade_samples", "ms_resolve", "ms_total")},
      "samples", st["shadow_samples"], "sum", int(u8.sum()))
