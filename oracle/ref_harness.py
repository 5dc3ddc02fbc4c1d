"""Drives the UNMODIFIED reference renderer as the parity oracle.

TEST INFRASTRUCTURE -- build container only.  Nothing here is imported by the
product (``raytracer_amd``); it exists to (a) generate the golden fixtures
under ``tests/golden/`` and (b) let CPU tests compare ``oracle/rt_oracle.c``
with the reference when the reference tree is present.  The reference's files
are never copied: they are imported from ``RT_REFERENCE_DIR`` (default
``/root/reference``) and a handful of module-level names are *wrapped* at run
time.  On a machine without that tree :func:`available` is False and the
callers skip.

What is wrapped, and why it works without editing the reference:

* ``Camera.get_ray`` (/root/reference/camera.py:18) -- records the pixel index
  ``y*W+x`` and resets the ray-tree node id to 1.
* ``ray_tracer.calculate_reflection_contribution`` / ``calculate_transparency_
  contribution`` (/root/reference/ray_tracer.py:174, :210) -- ``trace_ray``
  looks them up as module globals on every call, so rebinding the names makes
  node ids follow the recursion (reflection child 2k, transparency child 2k+1).
* ``Light.generate_samples`` (/root/reference/light.py:35) -- records which light
  is being sampled and restarts the draw counter.
* ``numpy.random.random`` -- returns the counter-based value of
  ``oracle/jitter.py`` instead of the Mersenne-Twister draw
  (/root/reference/light.py:53-54 are the only two call sites).
* ``ray_tracer.trace_ray``, ``calculate_phong_shading``,
  ``compute_shadow_ray_ratio_vectorized`` and
  ``SceneSettings.find_nearest_intersection`` -- counted / recorded, then
  forwarded unchanged.
"""
from __future__ import annotations

import contextlib
import os
import sys

import numpy as np

from . import jitter

REFERENCE_DIR = os.environ.get("RT_REFERENCE_DIR", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "ray_tracer.py"))


_mods = None


_REF_MODULES = ("utils", "material", "camera", "light", "scene_settings", "surfaces", "surfaces.sphere",
                "surfaces.infinite_plane", "surfaces.cube", "ray_tracer")


def load():
    """Import the reference's modules FROM SOURCE, under their own flat names, without leaving anything behind.

    The reference uses flat imports (``from camera import Camera``), so its names have to resolve while it loads.  A
    finder limited to exactly its module names serves them from ``REFERENCE_DIR`` for the duration of the load; the
    loader compiles the ``.py`` text and never reads the tree's ``__pycache__`` (bytecode shipped with an untrusted
    tree is not executed).  Afterwards the finder is removed, ``sys.path`` is untouched, and ``sys.modules`` gets back
    whatever it held under those names (this repository has a ``ray_tracer`` of its own).
    """
    global _mods
    if _mods is None:
        if not available():
            raise RuntimeError(f"reference tree not found at {REFERENCE_DIR}")
        import importlib
        import importlib.abc
        import importlib.machinery
        import importlib.util

        class SourceOnlyLoader(importlib.machinery.SourceFileLoader):
            def get_code(self, fullname):
                path = self.get_filename(fullname)
                with open(path, "rb") as f:
                    return compile(f.read(), path, "exec", dont_inherit=True)

        class Finder(importlib.abc.MetaPathFinder):
            def find_spec(self, fullname, path=None, target=None):
                if fullname not in _REF_MODULES:
                    return None
                rel = fullname.replace(".", os.sep)
                pkg = os.path.join(REFERENCE_DIR, rel, "__init__.py")
                if os.path.isfile(pkg):
                    return importlib.util.spec_from_file_location(fullname, pkg, loader=SourceOnlyLoader(fullname, pkg),
                                                                  submodule_search_locations=[os.path.dirname(pkg)])
                src = os.path.join(REFERENCE_DIR, rel + ".py")
                if os.path.isfile(src):
                    return importlib.util.spec_from_file_location(fullname, src, loader=SourceOnlyLoader(fullname, src))
                return None

        saved = {name: sys.modules.pop(name) for name in _REF_MODULES if name in sys.modules}
        finder = Finder()
        sys.meta_path.insert(0, finder)
        try:
            ref_rt = importlib.import_module("ray_tracer")
            loaded = {name: sys.modules[name] for name in _REF_MODULES if name in sys.modules}
        finally:
            sys.meta_path.remove(finder)
            for name in _REF_MODULES:
                sys.modules.pop(name, None)
            sys.modules.update(saved)
        for name, mod in loaded.items():
            assert os.path.abspath(mod.__file__).startswith(os.path.abspath(REFERENCE_DIR) + os.sep), (name, mod.__file__)
        _mods = {"rt": ref_rt, "camera": loaded["camera"], "light": loaded["light"], "scene": loaded["scene_settings"],
                 "sphere": loaded["surfaces.sphere"], "plane": loaded["surfaces.infinite_plane"], "cube": loaded["surfaces.cube"]}
    return _mods


class Probe:
    """State shared by the wrappers during one instrumented run."""

    def __init__(self, seed, record_hits=False):
        self.seed = seed
        self.record_hits = record_hits
        self.pixel = 0
        self.node = 1
        self.light = 0
        self.grid = 1
        self.draw = 0
        self.reset_counters()

    def reset_counters(self):
        self.primary = 0
        self.secondary = 0
        self.depth_cut = 0
        self.shaded = 0
        self.shadow = 0
        self.depth_hist = {}
        self.hits = []  # (pixel, node, surface index or -1, t, P[3], n[3])


@contextlib.contextmanager
def instrumented(scene, seed=0, counter_jitter=True, record_hits=False):
    """Context manager that installs the wrappers around the reference and removes them again."""
    m = load()
    rt, Camera, Light, Scene = m["rt"], m["camera"].Camera, m["light"].Light, m["scene"].SceneSettings
    probe = Probe(seed, record_hits)
    saved = {
        "get_ray": Camera.get_ray,
        "refl": rt.calculate_reflection_contribution,
        "trans": rt.calculate_transparency_contribution,
        "gen": Light.generate_samples,
        "rand": np.random.random,
        "trace": rt.trace_ray,
        "phong": rt.calculate_phong_shading,
        "shadow": rt.compute_shadow_ray_ratio_vectorized,
        "nearest": Scene.find_nearest_intersection,
    }
    max_rec = scene.max_recursions
    light_index = {id(l): i for i, l in enumerate(scene.lights)}
    surf_index = {id(s): i for i, s in enumerate(scene.surfaces)}

    def get_ray(self, px, py, w, h):
        probe.pixel = py * w + px
        probe.node = 1
        return saved["get_ray"](self, px, py, w, h)

    def refl(*a, **k):
        parent = probe.node
        probe.node = 2 * parent
        try:
            return saved["refl"](*a, **k)
        finally:
            probe.node = parent

    def trans(*a, **k):
        parent = probe.node
        probe.node = 2 * parent + 1
        try:
            return saved["trans"](*a, **k)
        finally:
            probe.node = parent

    def gen(self, right, up, n):
        probe.light = light_index[id(self)]
        probe.grid = int(n)
        probe.draw = 0
        return saved["gen"](self, right, up, n)

    def rand(*a, **k):
        if a or k:
            return saved["rand"](*a, **k)
        v = jitter.uniform(probe.seed, probe.pixel, probe.node, probe.light, probe.grid, probe.draw)
        probe.draw += 1
        return v

    def trace(o, d, sc, n, current_depth=0):
        probe.depth_hist[current_depth] = probe.depth_hist.get(current_depth, 0) + 1
        if current_depth >= max_rec:
            probe.depth_cut += 1
        elif current_depth == 0:
            probe.primary += 1
        else:
            probe.secondary += 1
        return saved["trace"](o, d, sc, n, current_depth)

    def phong(*a, **k):
        probe.shaded += 1
        return saved["phong"](*a, **k)

    def shadow(hit, light, sc, n):
        probe.shadow += 1 if (light.shadow_intensity == 0 or n == 1) else n * n
        return saved["shadow"](hit, light, sc, n)

    def nearest(self, o, d, *a, **k):
        t, p, nrm, s = saved["nearest"](self, o, d, *a, **k)
        if probe.record_hits:
            if s is None:
                probe.hits.append((probe.pixel, probe.node, -1, 0.0, (0.0,) * 3, (0.0,) * 3))
            else:
                probe.hits.append((probe.pixel, probe.node, surf_index[id(s)], float(t),
                                   tuple(float(v) for v in p), tuple(float(v) for v in nrm)))
        return t, p, nrm, s

    Camera.get_ray = get_ray
    rt.calculate_reflection_contribution = refl
    rt.calculate_transparency_contribution = trans
    rt.trace_ray = trace
    rt.calculate_phong_shading = phong
    rt.compute_shadow_ray_ratio_vectorized = shadow
    Scene.find_nearest_intersection = nearest
    if counter_jitter:
        Light.generate_samples = gen
        np.random.random = rand
    try:
        yield probe
    finally:
        Camera.get_ray = saved["get_ray"]
        rt.calculate_reflection_contribution = saved["refl"]
        rt.calculate_transparency_contribution = saved["trans"]
        rt.trace_ray = saved["trace"]
        rt.calculate_phong_shading = saved["phong"]
        rt.compute_shadow_ray_ratio_vectorized = saved["shadow"]
        Scene.find_nearest_intersection = saved["nearest"]
        Light.generate_samples = saved["gen"]
        np.random.random = saved["rand"]


def parse(path):
    """The reference's own parser (/root/reference/ray_tracer.py:371)."""
    return load()["rt"].parse_scene_file(path)


def render_pixels(scene, width, height, xs, ys, seed=0, record_hits=False):
    """Colour (already ``clip(c,0,1)*255``, float64) of the listed pixels, counter-based jitter.

    Mirrors what the body of the reference's frame loop does for one pixel
    (/root/reference/ray_tracer.py:319-331) but for an arbitrary subset, which the
    keyed stream makes equal to the full-frame value.
    Returns ``(rgb[n,3], counters dict, hits list)``.
    """
    rt = load()["rt"]
    n_shadow = int(scene.root_number_shadow_rays)
    out = np.zeros((len(xs), 3))
    with instrumented(scene, seed, True, record_hits) as probe:
        for k, (x, y) in enumerate(zip(xs, ys)):
            o, d = scene.camera.get_ray(int(x), int(y), width, height)
            c = rt.trace_ray(o, d, scene, n_shadow, current_depth=0)
            out[k] = np.clip(c, 0, 1) * 255
        counters = {
            "primary": probe.primary, "secondary": probe.secondary, "depth_cut": probe.depth_cut,
            "shaded": probe.shaded, "shadow": probe.shadow,
            "depth_hist": {int(k): int(v) for k, v in sorted(probe.depth_hist.items())},
        }
        hits = probe.hits
    return out, counters, hits


def render_rows(scene, width, height, rows, seed=0):
    """Full rows (list of y) with counter-based jitter -> (len(rows), width, 3) float64 + counters."""
    xs = np.tile(np.arange(width), len(rows))
    ys = np.repeat(np.asarray(rows), width)
    rgb, counters, _ = render_pixels(scene, width, height, xs, ys, seed)
    return rgb.reshape(len(rows), width, 3), counters


def render_frame_mt(scene, width, height, mt_seed):
    """The reference's own ``render()`` with its own Mersenne-Twister jitter (free-running mode)."""
    rt = load()["rt"]
    import io
    np.random.seed(mt_seed)
    with contextlib.redirect_stdout(io.StringIO()):
        return rt.render(scene, width, height)
