"""SceneSettings -> the plain arrays of ``include/rt_amd.h`` (ctypes mirror of the C structs).

Works on this package's records and, by duck typing, on the reference's own
objects (/root/reference/scene_settings.py, camera.py, light.py, material.py,
surfaces/*.py): only attribute names and the class names ``Sphere`` /
``InfinitePlane`` / ``Cube`` are used.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

RT_ABI_VERSION = 3
RT_SURF_SPHERE, RT_SURF_PLANE, RT_SURF_BOX, RT_SURF_PLANE_RAW = 0, 1, 2, 3
RT_MAT_STRIDE, RT_LIGHT_STRIDE = 11, 9


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("cam_pos", C.c_double * 3), ("cam_look_at", C.c_double * 3), ("cam_up", C.c_double * 3),
        ("cam_screen_dist", C.c_double), ("cam_screen_width", C.c_double),
        ("background", C.c_double * 3),
        ("n_shadow", C.c_int32), ("max_rec", C.c_int32),
        ("n_surfaces", C.c_int32), ("n_materials", C.c_int32), ("n_lights", C.c_int32), ("reserved", C.c_int32),
        ("surf_type", C.POINTER(C.c_int32)), ("surf_param", C.POINTER(C.c_double)),
        ("surf_material", C.POINTER(C.c_int32)),
        ("mat_param", C.POINTER(C.c_double)), ("light_param", C.POINTER(C.c_double)),
    ]


class RtRenderParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32),
        ("stripe_rows", C.c_int32), ("stripe_count", C.c_int32), ("stripe_index", C.c_int32),
        ("seed", C.c_uint32), ("flags", C.c_uint32),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64), ("secondary_rays", C.c_uint64), ("depth_cut", C.c_uint64),
        ("shaded_points", C.c_uint64), ("shadow_rays", C.c_uint64), ("rows_rendered", C.c_uint64),
        ("bands", C.c_uint32), ("reserved", C.c_uint32),
        ("ms_total", C.c_double), ("ms_trace", C.c_double), ("ms_shade", C.c_double),
        ("ms_resolve", C.c_double), ("flop_shade", C.c_double), ("wave_light", C.c_uint64 * 8),
        ("ms_shade_main", C.c_double), ("shadow_samples", C.c_uint64),
    ]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_ if name not in ("reserved", "wave_light")}
        d["wave_light"] = list(self.wave_light)
        return d

    @property
    def total_rays(self):
        return self.primary_rays + self.secondary_rays + self.shadow_rays


class PackedScene:
    """Owns the NumPy arrays an ``RtSceneDesc`` points into (keep it alive during the call)."""

    def __init__(self, desc, arrays):
        self.desc = desc
        self._arrays = arrays

    @property
    def n_surfaces(self):
        return self.desc.n_surfaces


def _material_slot(material_index, n_materials):
    """0-based slot of a 1-based ``material_index``.

    The reference addresses ``materials[material_index - 1]``
    (/root/reference/ray_tracer.py:272), so index 0 silently means the LAST material
    (Python's ``[-1]``); that is kept.  Anything outside ``[-M+1, M]`` would raise
    IndexError there when the surface is first hit; here it is rejected up front.
    """
    k = int(material_index) - 1
    if k < 0:
        k += n_materials
    if not 0 <= k < n_materials:
        raise IndexError(f"material_index {material_index} out of range for {n_materials} materials")
    return k


def effective_depth(max_recursions):
    """Depth levels that are actually traced: ``depth >= max_recursions`` cuts (ray_tracer.py:261)."""
    return max(0, int(math.ceil(float(max_recursions))))


def pack_scene(scene) -> PackedScene:
    cam = scene.camera
    if cam is None:
        raise ValueError("scene has no camera ('cam' line missing)")
    if scene.background_color is None or scene.root_number_shadow_rays is None or scene.max_recursions is None:
        raise ValueError("scene has no settings ('set' line missing)")
    n_shadow = int(scene.root_number_shadow_rays)  # ray_tracer.py:312
    if n_shadow < 1:
        raise ValueError(f"root_number_shadow_rays must be >= 1 (got {scene.root_number_shadow_rays}); "
                         "the reference produces NaN colours for it")
    S, M, L = len(scene.surfaces), len(scene.materials), len(scene.lights)
    surf_type = np.zeros(max(S, 1), dtype=np.int32)
    surf_param = np.zeros((4, max(S, 1)), dtype=np.float64)
    surf_mat = np.zeros(max(S, 1), dtype=np.int32)
    for i, s in enumerate(scene.surfaces):
        kind = type(s).__name__
        if kind == "Sphere":
            surf_type[i] = RT_SURF_SPHERE
            surf_param[0:3, i] = np.asarray(s.position, dtype=np.float64)
            surf_param[3, i] = float(s.radius)
        elif kind == "InfinitePlane":
            raw = getattr(s, "raw_normal", None)
            if raw is not None:
                surf_type[i] = RT_SURF_PLANE_RAW
                surf_param[0:3, i] = np.asarray(raw, dtype=np.float64)
            else:
                surf_type[i] = RT_SURF_PLANE
                surf_param[0:3, i] = np.asarray(s.normal, dtype=np.float64)
            surf_param[3, i] = float(s.offset)
        elif kind == "Cube":
            surf_type[i] = RT_SURF_BOX
            surf_param[0:3, i] = np.asarray(s.position, dtype=np.float64)
            surf_param[3, i] = float(s.scale)
        else:
            raise TypeError(f"unsupported surface type {kind!r}")
        surf_mat[i] = _material_slot(s.material_index, M)
    mat = np.zeros((RT_MAT_STRIDE, max(M, 1)), dtype=np.float64)
    for i, m in enumerate(scene.materials):
        mat[0:3, i] = m.diffuse_color
        mat[3:6, i] = m.specular_color
        mat[6:9, i] = m.reflection_color
        mat[9, i] = m.shininess
        mat[10, i] = m.transparency
    lgt = np.zeros((RT_LIGHT_STRIDE, max(L, 1)), dtype=np.float64)
    for i, l in enumerate(scene.lights):
        lgt[0:3, i] = l.position
        lgt[3:6, i] = l.color
        lgt[6, i] = l.specular_intensity
        lgt[7, i] = l.shadow_intensity
        lgt[8, i] = l.radius

    d = RtSceneDesc()
    d.struct_size = C.sizeof(RtSceneDesc)
    d.abi_version = RT_ABI_VERSION
    d.cam_pos[:] = [float(v) for v in cam.position]
    d.cam_look_at[:] = [float(v) for v in cam.look_at]
    d.cam_up[:] = [float(v) for v in cam.up_vector]
    d.cam_screen_dist = float(cam.screen_distance)
    d.cam_screen_width = float(cam.screen_width)
    d.background[:] = [float(v) for v in scene.background_color]
    d.n_shadow = n_shadow
    d.max_rec = effective_depth(scene.max_recursions)
    d.n_surfaces, d.n_materials, d.n_lights = S, M, L
    d.surf_type = surf_type.ctypes.data_as(C.POINTER(C.c_int32))
    d.surf_param = surf_param.ctypes.data_as(C.POINTER(C.c_double))
    d.surf_material = surf_mat.ctypes.data_as(C.POINTER(C.c_int32))
    d.mat_param = mat.ctypes.data_as(C.POINTER(C.c_double))
    d.light_param = lgt.ctypes.data_as(C.POINTER(C.c_double))
    return PackedScene(d, (surf_type, surf_param, surf_mat, mat, lgt))


def render_params(width, height, row_begin=0, row_end=None, stripe_rows=1, stripe_count=1, stripe_index=0,
                  seed=0, flags=0) -> RtRenderParams:
    p = RtRenderParams()
    p.struct_size = C.sizeof(RtRenderParams)
    p.width, p.height = int(width), int(height)
    p.row_begin = int(row_begin)
    p.row_end = int(height if row_end is None else row_end)
    p.stripe_rows, p.stripe_count, p.stripe_index = int(stripe_rows), int(stripe_count), int(stripe_index)
    p.seed = int(seed) & 0xFFFFFFFF
    p.flags = int(flags)
    return p


def selected_rows(width, height, row_begin=0, row_end=None, stripe_rows=1, stripe_count=1, stripe_index=0):
    """Rows a (row range, stripe) selection covers, in increasing y (host-side mirror of rt_rows_selected)."""
    row_end = height if row_end is None else row_end
    return [y for y in range(row_begin, row_end) if (y // stripe_rows) % stripe_count == stripe_index]
