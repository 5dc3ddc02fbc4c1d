"""MI355X-native renderer for the per-pixel frame loop of roiguri/raytracer.

Host side (this package): the reference's scene model, file format and
``render()`` signature.  Device side: ``librt_amd.so`` -- hand-written HIP kernels
for gfx950 behind the C ABI of ``include/rt_amd.h``.
"""
from .ray_tracer import parse_scene_file, render  # noqa: F401

__all__ = ["parse_scene_file", "render"]
