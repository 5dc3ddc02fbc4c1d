"""ctypes front end of ``oracle/rt_oracle.c`` (TEST INFRASTRUCTURE, see that file's header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import packing

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "librt_oracle.so")


class RtoHit(C.Structure):
    _fields_ = [("pixel", C.c_uint32), ("node", C.c_uint32), ("surf", C.c_int32), ("pad", C.c_int32),
                ("t", C.c_double), ("P", C.c_double * 3), ("n", C.c_double * 3)]


def build(force=False):
    """Compile the C restatement with gcc (a checker is being built, not used)."""
    src = os.path.join(_HERE, "rt_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.rto_render_pixels.restype = C.c_int
        L.rto_render_pixels.argtypes = [C.POINTER(packing.RtSceneDesc), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.rto_camera_basis.restype = C.c_int
        L.rto_camera_basis.argtypes = [C.POINTER(packing.RtSceneDesc), C.c_void_p]
        L.rto_intersect_scalar.restype = C.c_int
        L.rto_intersect_scalar.argtypes = [C.c_int] + [C.c_void_p] * 5
        L.rto_intersect_batch.restype = C.c_double
        L.rto_intersect_batch.argtypes = [C.c_int] + [C.c_void_p] * 3
        L.rto_jitter_bits.restype = None
        L.rto_jitter_bits.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)] * 2
        _lib = L
    return _lib


COUNTER_NAMES = ("primary", "secondary", "depth_cut", "shaded", "shadow")


def render_pixels(scene, width, height, xs, ys, seed=0, threads=1, record_hits=False):
    """Colours (clip*255, float64) of the listed pixels -> (rgb[n,3], counters dict, hits or None)."""
    packed = scene if isinstance(scene, packing.PackedScene) else packing.pack_scene(scene)
    xs = np.ascontiguousarray(xs, dtype=np.int32)
    ys = np.ascontiguousarray(ys, dtype=np.int32)
    n = len(xs)
    out = np.zeros((n, 3), dtype=np.float64)
    cnt = np.zeros(5, dtype=np.uint64)
    hits = None
    nh = C.c_int64(0)
    cap = 0
    if record_hits:
        cap = max(1024, n * 64)
        hits = (RtoHit * cap)()
    rc = lib().rto_render_pixels(C.byref(packed.desc), int(width), int(height), xs.ctypes.data, ys.ctypes.data, n,
                                 int(seed) & 0xFFFFFFFF, int(threads), out.ctypes.data, cnt.ctypes.data,
                                 C.cast(hits, C.c_void_p) if hits is not None else None, cap, C.byref(nh))
    if rc != 0:
        raise ValueError("oracle rejected the scene descriptor")
    counters = {k: int(v) for k, v in zip(COUNTER_NAMES, cnt)}
    hit_list = None
    if record_hits:
        hit_list = [(h.pixel, h.node, h.surf, h.t, tuple(h.P), tuple(h.n)) for h in hits[: nh.value]]
    return out, counters, hit_list


def render_rows(scene, width, height, rows, seed=0, threads=1):
    rows = list(rows)
    xs = np.tile(np.arange(width, dtype=np.int32), len(rows))
    ys = np.repeat(np.asarray(rows, dtype=np.int32), width)
    rgb, counters, _ = render_pixels(scene, width, height, xs, ys, seed, threads)
    return rgb.reshape(len(rows), width, 3), counters


def render_frame(scene, width, height, seed=0, threads=1):
    return render_rows(scene, width, height, range(height), seed, threads)


def camera_basis(scene):
    packed = scene if isinstance(scene, packing.PackedScene) else packing.pack_scene(scene)
    out = np.zeros(9)
    if lib().rto_camera_basis(C.byref(packed.desc), out.ctypes.data) != 0:
        raise ValueError("oracle rejected the scene descriptor")
    return out.reshape(3, 3)


def intersect_scalar(kind, params, o, d):
    p = np.ascontiguousarray(params, dtype=np.float64)
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    t = C.c_double(0)
    n = np.zeros(3)
    hit = lib().rto_intersect_scalar(int(kind), p.ctypes.data, o.ctypes.data, d.ctypes.data, C.addressof(t), n.ctypes.data)
    return (t.value, n) if hit == 1 else (None, None)


def intersect_batch(kind, params, o, d):
    p = np.ascontiguousarray(params, dtype=np.float64)
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    return lib().rto_intersect_batch(int(kind), p.ctypes.data, o.ctypes.data, d.ctypes.data)


def jitter_bits(seed, pixel, node, light, n, s):
    u, v = C.c_uint32(0), C.c_uint32(0)
    lib().rto_jitter_bits(seed, pixel, node, light, n, s, C.byref(u), C.byref(v))
    return u.value, v.value
