#!/usr/bin/env python3
"""Development helper: reads the per-phase cycle counters of the shading kernel after rendering a scene.  The counters
are not in the shipped source: apply tools/patches/phase_timing.patch (git apply), build the variant with
tools/build_variant.sh phase -DRT_EXP_PHASE, revert the patch, then
    RT_AMD_LIB=$PWD/raytracer_amd/librt_phase.so python tools/exp_phase.py pool_n16 3840 2160
(result of this round: profiles/r03_phase_timing.txt)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R  # noqa: E402
from raytracer_amd import runtime  # noqa: E402

name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = runtime.load_library()
rd = lib.rt_exp_phase_read
rd.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
r = runtime.Renderer(0)
r.set_scene(R.parse_scene_file(os.path.join(ROOT, "scenes", name + ".txt")))
buf = (ctypes.c_ulonglong * 16)()
for it in range(3):
    _, _, stats = r.render_rows(w, h, seed=it)
    assert rd(buf) == 0
    v = list(buf)
    waves = v[12]
    tot = sum(v[:10])
    names = ["0 work fetch", "1 chunk load + prefilter", "2 light load, direction", "3 light basis, phong test", "4 surface classification",
             "5 own sphere, exact list, defer", "6 key, box rows", "7 sample loop", "8 phong + sum", "9 atomics"]
    print(f"frame {it}: {waves} waves, {tot / max(waves, 1):.0f} clock ticks per wave; ms_shade {stats.get('ms_shade')}")
    for i, n in enumerate(names):
        print(f"   {n:34s} {v[i] / max(waves, 1):12.0f}  {100.0 * v[i] / max(tot, 1):5.1f} %")
r.close()
