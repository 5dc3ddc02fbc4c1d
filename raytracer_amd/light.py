"""Light record (mirrors the constructor of /root/reference/light.py:6-11).

The per-hit sampling basis and the N x N jittered sample grid
(/root/reference/light.py:13-65) are generated inside the shading kernel.
"""


class Light:
    __slots__ = ("position", "color", "specular_intensity", "shadow_intensity", "radius")

    def __init__(self, position, color, specular_intensity, shadow_intensity, radius):
        self.position = list(position)
        self.color = list(color)
        self.specular_intensity = specular_intensity
        self.shadow_intensity = shadow_intensity
        self.radius = radius

    def __repr__(self):
        return (f"Light(pos={self.position}, color={self.color}, spec={self.specular_intensity}, "
                f"shadow={self.shadow_intensity}, radius={self.radius})")
