"""Sphere record (constructor of /root/reference/surfaces/sphere.py:6-9); intersection runs on the GPU."""
import numpy as np


class Sphere:
    def __init__(self, position, radius, material_index):
        self.position = np.array(position, dtype=np.float64)
        self.radius = radius
        self.material_index = material_index
