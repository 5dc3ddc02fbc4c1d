"""ctypes binding of ``librt_amd.so`` (C ABI: ``include/rt_amd.h``) and the per-GPU ``Renderer``.

There is deliberately no fallback: if the shared library has not been built
(``python -m raytracer_amd.build``) or no GPU is visible, loading or creating a
renderer raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import packing

# RT_AMD_LIB names another build of the same library (A/B measurements of two builds in one session)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librt_amd.so")

EXPORTS = ("rt_create", "rt_destroy", "rt_last_error", "rt_set_scene", "rt_render", "rt_render_device",
           "rt_render_device_noalloc", "rt_render_enqueue", "rt_render_wait", "rt_deliver_rows", "rt_rows_selected", "rt_debug_hits", "rt_jitter_bits")

HF_CBATCH_POS, HF_CDOT_POS, HF_BOX_OUTSIDE, HF_NO_OWN = 1, 2, 4, 8


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"librt_amd error {code}: {message}")
        self.code = code


_lib = None


def load_library():
    """Load the HIP library; raises if it is missing (the product path has no CPU route)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m raytracer_amd.build` "
                           "(the renderer has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_create.restype = C.c_int
    L.rt_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_destroy.restype = None
    L.rt_destroy.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_last_error.argtypes = [vp]
    L.rt_set_scene.restype = C.c_int
    L.rt_set_scene.argtypes = [vp, C.POINTER(packing.RtSceneDesc)]
    L.rt_render.restype = C.c_int
    L.rt_render.argtypes = [vp, C.POINTER(packing.RtRenderParams), vp, vp, C.POINTER(packing.RtStats)]
    L.rt_render_device.restype = C.c_int
    L.rt_render_device.argtypes = [vp, C.POINTER(packing.RtRenderParams), vp, vp, vp, C.POINTER(packing.RtStats)]
    L.rt_render_device_noalloc.restype = C.c_int
    L.rt_render_device_noalloc.argtypes = [vp, C.POINTER(packing.RtRenderParams), vp, vp, vp, C.POINTER(packing.RtStats)]
    L.rt_render_enqueue.restype = C.c_int
    L.rt_render_enqueue.argtypes = [vp, C.POINTER(packing.RtRenderParams), vp, vp, vp]
    L.rt_render_wait.restype = C.c_int
    L.rt_render_wait.argtypes = [vp, C.POINTER(packing.RtStats)]
    L.rt_deliver_rows.restype = C.c_int
    L.rt_deliver_rows.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int, vp]
    L.rt_rows_selected.restype = C.c_int
    L.rt_rows_selected.argtypes = [C.POINTER(packing.RtRenderParams)]
    L.rt_debug_hits.restype = C.c_int
    L.rt_debug_hits.argtypes = [vp, C.POINTER(packing.RtRenderParams), C.c_uint32] + [vp] * 7 + [C.POINTER(C.c_uint32)]
    L.rt_jitter_bits.restype = None
    L.rt_jitter_bits.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)] * 2
    _lib = L
    return L


def jitter_bits(seed, pixel, node, light, n, s):
    u, v = C.c_uint32(0), C.c_uint32(0)
    load_library().rt_jitter_bits(seed & 0xFFFFFFFF, pixel, node, light, n, s, C.byref(u), C.byref(v))
    return u.value, v.value


def _stats_dict(st: packing.RtStats):
    d = st.as_dict()
    d["total_rays"] = st.total_rays
    return d


class Renderer:
    """One GPU context (one ``rt_handle``): upload a scene once, render any number of frames."""

    _scene_serial = 0

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.rt_create(int(device), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise RtError(rc, f"rt_create(device={device}) failed -- is a GPU visible?")
        self.device = int(device)
        self._scene_token = None
        self._packed = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.rt_last_error(self._h)
            raise RtError(rc, msg.decode() if msg else "")

    def set_scene(self, scene):
        """Upload a SceneSettings (or an already packed scene)."""
        packed = scene if isinstance(scene, packing.PackedScene) else packing.pack_scene(scene)
        self._check(self._lib.rt_set_scene(self._h, C.byref(packed.desc)))
        self._packed = packed
        Renderer._scene_serial += 1
        self._scene_token = Renderer._scene_serial     # changes with every upload (frames.FrameStream sizes its queues per scene)
        return self

    def render_rows(self, width, height, *, seed=0, row_begin=0, row_end=None, stripe_rows=1, stripe_count=1,
                    stripe_index=0, want_f32=True, want_u8=True, flags=0):
        """Render the selected rows of a frame; returns ``(f32 or None, u8 or None, stats dict)``."""
        p = packing.render_params(width, height, row_begin, row_end, stripe_rows, stripe_count, stripe_index, seed, flags)
        rows = self._lib.rt_rows_selected(C.byref(p))
        f32 = np.empty((rows, int(width), 3), dtype=np.float32) if want_f32 else None
        u8 = np.empty((rows, int(width), 3), dtype=np.uint8) if want_u8 else None
        st = packing.RtStats()
        self._check(self._lib.rt_render(self._h, C.byref(p), f32.ctypes.data if want_f32 else None,
                                        u8.ctypes.data if want_u8 else None, C.byref(st)))
        return f32, u8, _stats_dict(st)

    def render_rows_device(self, width, height, dev_f32_ptr, dev_u8_ptr, *, stream=None, seed=0, row_begin=0,
                           row_end=None, stripe_rows=1, stripe_count=1, stripe_index=0, flags=0):
        """Render into DEVICE buffers (raw pointers, e.g. ``tensor.data_ptr()``); returns the stats dict."""
        p = packing.render_params(width, height, row_begin, row_end, stripe_rows, stripe_count, stripe_index, seed, flags)
        st = packing.RtStats()
        self._check(self._lib.rt_render_device(self._h, C.byref(p), dev_f32_ptr or None, dev_u8_ptr or None,
                                               stream or None, C.byref(st)))
        return _stats_dict(st)

    def render_rows_device_noalloc(self, width, height, dev_f32_ptr, dev_u8_ptr, *, stream=None, seed=0, row_begin=0,
                                   row_end=None, stripe_rows=1, stripe_count=1, stripe_index=0, flags=0):
        """``render_rows_device`` without any device allocation: the queues live in the library's static scratch array
        (at most ``RT_NOALLOC_MAX_PIXELS`` pixels per call; calls that overlap on one GPU race)."""
        p = packing.render_params(width, height, row_begin, row_end, stripe_rows, stripe_count, stripe_index, seed, flags)
        st = packing.RtStats()
        self._check(self._lib.rt_render_device_noalloc(self._h, C.byref(p), dev_f32_ptr or None, dev_u8_ptr or None,
                                                       stream or None, C.byref(st)))
        return _stats_dict(st)

    def enqueue_rows_device(self, width, height, dev_f32_ptr, dev_u8_ptr, *, stream=None, seed=0, row_begin=0,
                            row_end=None, stripe_rows=1, stripe_count=1, stripe_index=0, flags=0):
        """Like ``render_rows_device`` but returns as soon as the frame is on the stream; ``wait()`` collects."""
        p = packing.render_params(width, height, row_begin, row_end, stripe_rows, stripe_count, stripe_index, seed, flags)
        self._check(self._lib.rt_render_enqueue(self._h, C.byref(p), dev_f32_ptr or None, dev_u8_ptr or None, stream or None))

    def deliver_rows(self, host_ptr, dev_ptr, rows, row_bytes, *, src_row_ptr=None, workgroups=0, stream=None):
        """Enqueue the copy of ``rows`` rows of ``row_bytes`` bytes from device memory into PINNED host memory on
        ``stream`` (raw pointers); ``src_row_ptr``: device int32 array naming the source row of every destination row;
        ``workgroups``: footprint of the copy kernel (0 = 8: hides beside other kernels; 64 reaches the PCIe rate)."""
        self._check(self._lib.rt_deliver_rows(self._h, host_ptr, dev_ptr, src_row_ptr or None, int(rows), int(row_bytes),
                                              int(workgroups), stream or None))

    def wait(self):
        """Block until every enqueued frame is done; returns the summed stats dict (raises if one overflowed)."""
        st = packing.RtStats()
        self._check(self._lib.rt_render_wait(self._h, C.byref(st)))
        return _stats_dict(st)

    def render(self, scene, width, height, *, seed=0):
        """Full frame of ``scene`` -> ``(float32 (H,W,3) in [0,255], stats dict)``."""
        if scene is not None:
            self.set_scene(scene)
        f32, _, stats = self.render_rows(width, height, seed=seed, want_u8=False)
        return f32, stats

    def debug_hits(self, width, height, *, row_begin=0, row_end=None, capacity=None):
        """Float64 record of every nearest-hit query of the selected rows (parity probe)."""
        p = packing.render_params(width, height, row_begin, row_end)
        rows = self._lib.rt_rows_selected(C.byref(p))
        cap = int(capacity or max(1024, rows * int(width) * 16))
        pixel = np.zeros(cap, np.uint32); node = np.zeros(cap, np.uint32); surf = np.zeros(cap, np.int32)
        t = np.zeros(cap); P = np.zeros((cap, 3)); n = np.zeros((cap, 3)); flags = np.zeros(cap, np.uint32)
        cnt = C.c_uint32(0)
        self._check(self._lib.rt_debug_hits(self._h, C.byref(p), cap, pixel.ctypes.data, node.ctypes.data, surf.ctypes.data,
                                            t.ctypes.data, P.ctypes.data, n.ctypes.data, flags.ctypes.data, C.byref(cnt)))
        m = min(cnt.value, cap)
        return {"count": cnt.value, "pixel": pixel[:m], "node": node[:m], "surf": surf[:m], "t": t[:m], "P": P[:m],
                "n": n[:m], "flags": flags[:m]}


_renderers = {}


def default_renderer(device=None) -> Renderer:
    """Process-wide renderer of a device (created on first use)."""
    dev = 0 if device is None else int(device)
    r = _renderers.get(dev)
    if r is None:
        r = _renderers[dev] = Renderer(dev)
    return r
