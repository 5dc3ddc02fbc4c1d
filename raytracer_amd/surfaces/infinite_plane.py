"""Plane record (constructor of /root/reference/surfaces/infinite_plane.py:6-10).

``normal`` is the unit normal as the reference stores it; ``offset`` is kept as
written (the reference does not rescale it, infinite_plane.py:9).  ``raw_normal``
keeps the vector as written so the renderer can normalise it with pinned rounding.
"""
import numpy as np


class InfinitePlane:
    def __init__(self, normal, offset, material_index):
        self.raw_normal = np.array(normal, dtype=np.float64)
        self.normal = self.raw_normal / np.linalg.norm(self.raw_normal)
        self.offset = offset
        self.material_index = material_index
