"""Scene container (mirrors /root/reference/scene_settings.py:6-17).

``surfaces`` keeps file order: the nearest-hit search resolves ties in favour of
the earlier surface (/root/reference/scene_settings.py:35-40) and the kernels do
the same.  The brute-force query itself (scene_settings.py:19-46) runs on the GPU.
"""


class SceneSettings:
    def __init__(self, background_color, root_number_shadow_rays, max_recursions,
                 camera=None, surfaces=None, materials=None, lights=None):
        self.background_color = background_color
        self.root_number_shadow_rays = root_number_shadow_rays
        self.max_recursions = max_recursions
        self.camera = camera
        self.surfaces = [] if surfaces is None else surfaces
        self.materials = [] if materials is None else materials
        self.lights = [] if lights is None else lights
