"""Material record (mirrors the constructor of /root/reference/material.py:2-7).

Only the data lives on the host; the blend of local, reflected and transmitted
colour (/root/reference/material.py:26-30) is evaluated on the GPU.
"""


class Material:
    __slots__ = ("diffuse_color", "specular_color", "reflection_color", "shininess", "transparency")

    def __init__(self, diffuse_color, specular_color, reflection_color, shininess, transparency):
        self.diffuse_color = list(diffuse_color)
        self.specular_color = list(specular_color)
        self.reflection_color = list(reflection_color)
        self.shininess = shininess
        self.transparency = transparency

    def __repr__(self):
        return (f"Material(kd={self.diffuse_color}, ks={self.specular_color}, kr={self.reflection_color}, "
                f"shininess={self.shininess}, transparency={self.transparency})")
