"""Axis-aligned cube record (constructor of /root/reference/surfaces/cube.py:6-9); intersection runs on the GPU."""
import numpy as np


class Cube:
    def __init__(self, position, scale, material_index):
        self.position = np.array(position, dtype=np.float64)
        self.scale = scale
        self.material_index = material_index
