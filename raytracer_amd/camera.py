"""Camera record (mirrors the constructor of /root/reference/camera.py:6-16).

The renderer derives the orthonormal basis and the per-pixel rays
(/root/reference/camera.py:14-16, :18-44) itself, in float64 with the reference's
rounding, from ``position``, ``look_at`` and ``up_vector``; ``forward``/``right``/
``up`` are exposed here only for callers that want to inspect them.
"""
import numpy as np


def _unit(v):
    n = np.linalg.norm(v)
    return v if n == 0 else v / n


class Camera:
    def __init__(self, position, look_at, up_vector, screen_distance, screen_width):
        self.position = np.array(position, dtype=np.float64)
        self.look_at = np.array(look_at, dtype=np.float64)
        self.up_vector = np.array(up_vector, dtype=np.float64)
        self.screen_distance = screen_distance
        self.screen_width = screen_width

    @property
    def forward(self):
        return _unit(self.look_at - self.position)

    @property
    def right(self):
        return _unit(np.cross(self.forward, self.up_vector))

    @property
    def up(self):
        r = self.right
        return _unit(np.cross(r, self.forward))
