"""Command line, scene-file parser and ``render()`` of the MI355X renderer.

Same command line and file format as the reference
(/root/reference/ray_tracer.py:371-440):

    python ray_tracer.py <scene.txt> <out.png> [--width W] [--height H]

``render(scene_settings, width, height)`` keeps the reference's signature and
return contract (/root/reference/ray_tracer.py:297-368: an ``(H, W, 3)`` float64
array already clipped to [0, 255], row 0 on top) but its body hands the packed
scene to ``librt_amd.so`` (hand-written HIP for gfx950) instead of looping over
pixels in Python.  There is no CPU fallback: without the library or without a
GPU it raises.
"""
from __future__ import annotations

import argparse
import time

import numpy as np

from .camera import Camera
from .light import Light
from .material import Material
from .scene_settings import SceneSettings
from .surfaces.cube import Cube
from .surfaces.infinite_plane import InfinitePlane
from .surfaces.sphere import Sphere

_ARITY = {"cam": 11, "set": 5, "mtl": 11, "sph": 5, "pln": 5, "box": 5, "lgt": 9}


def parse_scene_file(file_path):
    """Text scene -> SceneSettings (format: /root/reference/ray_tracer.py:378-408).

    One object per line, ``keyword`` followed by floats; blank lines and lines
    starting with ``#`` are skipped; the last ``cam``/``set`` line wins; surfaces,
    materials and lights keep file order.  An unknown keyword raises ValueError
    like the reference; a line with too few numbers raises ValueError naming the
    line (the reference raises a bare IndexError there).
    """
    camera = None
    background = shadow_rays = max_rec = None
    surfaces, materials, lights = [], [], []
    with open(file_path, "r") as f:
        for lineno, raw in enumerate(f, 1):
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            keyword, *rest = line.split()
            if keyword not in _ARITY:
                raise ValueError("Unknown object type: {}".format(keyword))
            v = [float(tok) for tok in rest]
            if len(v) < _ARITY[keyword]:
                raise ValueError(f"{file_path}:{lineno}: '{keyword}' needs {_ARITY[keyword]} numbers, got {len(v)}")
            if keyword == "cam":
                camera = Camera(v[0:3], v[3:6], v[6:9], v[9], v[10])
            elif keyword == "set":
                background, shadow_rays, max_rec = v[0:3], v[3], v[4]
            elif keyword == "mtl":
                materials.append(Material(v[0:3], v[3:6], v[6:9], v[9], v[10]))
            elif keyword == "sph":
                surfaces.append(Sphere(v[0:3], v[3], int(v[4])))
            elif keyword == "pln":
                surfaces.append(InfinitePlane(v[0:3], v[3], int(v[4])))
            elif keyword == "box":
                surfaces.append(Cube(v[0:3], v[3], int(v[4])))
            elif keyword == "lgt":
                lights.append(Light(v[0:3], v[3:6], v[6], v[7], v[8]))
    return SceneSettings(background, shadow_rays, max_rec, camera, surfaces, materials, lights)


def render(scene_settings, width, height, *, seed=0, device=None, verbose=True, return_stats=False):
    """Render the scene on the GPU; drop-in for /root/reference/ray_tracer.py:297.

    ``seed`` selects the shadow-jitter stream (the reference draws from NumPy's
    global generator instead; pass a different seed per frame for its free-running
    behaviour).  Returns the image, or ``(image, stats dict)`` with ``return_stats``.
    """
    from . import runtime

    start = time.time()
    image, stats = runtime.default_renderer(device).render(scene_settings, int(width), int(height), seed=seed)
    elapsed = time.time() - start
    if verbose:
        mrays = stats["total_rays"] / max(stats["ms_total"], 1e-9) / 1e3
        print(f"  GPU: {stats['total_rays']} rays ({stats['primary_rays']} primary, {stats['secondary_rays']} secondary, "
              f"{stats['shadow_rays']} shadow) in {stats['ms_total']:.2f} ms = {mrays:.1f} Mrays/s")
        print(f"  Render time: {elapsed:.2f} seconds ({elapsed/60:.2f} minutes)")
    image = image.astype(np.float64)
    return (image, stats) if return_stats else image


def main(argv=None):
    from PIL import Image

    parser = argparse.ArgumentParser(description="Python Ray Tracer")
    parser.add_argument("scene_file", type=str, help="Path to the scene file")
    parser.add_argument("output_image", type=str, help="Name of the output image file")
    parser.add_argument("--width", type=int, default=500, help="Image width")
    parser.add_argument("--height", type=int, default=500, help="Image height")
    parser.add_argument("--seed", type=int, default=0, help="Shadow-jitter stream seed (extension)")
    args = parser.parse_args(argv)

    scene_settings = parse_scene_file(args.scene_file)
    image_array = render(scene_settings, args.width, args.height, seed=args.seed)

    Image.fromarray(np.uint8(image_array)).save(args.output_image)
    print(f"Image saved to {args.output_image}")


if __name__ == "__main__":
    main()
