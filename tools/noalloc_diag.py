"""development probe (GPU box): repeatability of rt_render_device / rt_render_device_noalloc on one frame"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import scene
from raytracer_amd import runtime

r = runtime.Renderer(0)
r.set_scene(scene(sys.argv[1] if len(sys.argv) > 1 else "pool"))
dev = torch.device("cuda", 0)
def run(call, w, h, seed):
    u8 = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
    f32 = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    st = call(w, h, f32.data_ptr(), u8.data_ptr(), seed=seed)
    torch.cuda.synchronize()
    return u8.cpu().numpy(), f32.cpu().numpy(), st
run(r.render_rows_device_noalloc, 320, 200, 3)
ref = run(r.render_rows_device, 96, 54, 4)
for k in range(6):
    for name, call in (("noalloc", r.render_rows_device_noalloc), ("device", r.render_rows_device)):
        if k % 2 == 0 and name == "noalloc":
            run(r.render_rows_device_noalloc, 320, 200, 3)
        u8, f32, st = run(call, 96, 54, 4)
        d = np.argwhere(u8 != ref[0])
        print(k, name, "differing bytes", len(d), [(tuple(x), int(u8[tuple(x)]), int(ref[0][tuple(x)]), float(f32[tuple(x)]), float(ref[1][tuple(x)])) for x in d[:6]],
              st["shaded_points"], st["wave_light"], flush=True)
