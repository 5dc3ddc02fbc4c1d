"""Scene-file parser and SoA packer (host side of the drop-in boundary)."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as R
from helpers import scene, write_scene
from raytracer_amd import packing
from raytracer_amd.surfaces.cube import Cube
from raytracer_amd.surfaces.infinite_plane import InfinitePlane
from raytracer_amd.surfaces.sphere import Sphere

MINI = """
# comment line
cam 0 0 5  0 0 0  0 1 0  1 2

set 0.1 0.2 0.3 3 4
mtl 1 0 0  1 1 1  0 0 0  10 0
mtl 0 1 0  1 1 1  0.5 0.5 0.5  20 0.25
box 1 2 3 2 1
pln 0 2 0 -1 2
sph 0 0 0 1.5 1
lgt 0 5 0  1 1 1  1 0.5 2
lgt 1 5 0  1 1 1  1 0 2
"""


def test_parse_keeps_file_order_and_types(tmp_path):
    s = R.parse_scene_file(write_scene(tmp_path, MINI))
    assert [type(x).__name__ for x in s.surfaces] == ["Cube", "InfinitePlane", "Sphere"]
    assert s.background_color == [0.1, 0.2, 0.3]
    assert s.root_number_shadow_rays == 3.0 and s.max_recursions == 4.0  # stay floats (ray_tracer.py:390-391)
    assert len(s.materials) == 2 and len(s.lights) == 2
    assert isinstance(s.surfaces[0], Cube) and s.surfaces[0].scale == 2.0 and s.surfaces[0].material_index == 1
    assert isinstance(s.surfaces[1], InfinitePlane) and s.surfaces[1].offset == -1.0
    np.testing.assert_array_equal(s.surfaces[1].normal, [0.0, 1.0, 0.0])       # normalised, offset NOT rescaled
    np.testing.assert_array_equal(s.surfaces[1].raw_normal, [0.0, 2.0, 0.0])
    assert isinstance(s.surfaces[2], Sphere) and s.surfaces[2].radius == 1.5
    assert s.lights[0].position == [0.0, 5.0, 0.0] and s.lights[0].radius == 2.0
    assert s.materials[1].transparency == 0.25 and s.materials[1].reflection_color == [0.5, 0.5, 0.5]
    np.testing.assert_array_equal(s.camera.position, [0, 0, 5])
    assert s.camera.screen_distance == 1.0 and s.camera.screen_width == 2.0


def test_last_cam_and_set_line_win(tmp_path):
    s = R.parse_scene_file(write_scene(tmp_path, MINI + "cam 1 1 1  0 0 0  0 1 0  2 3\nset 1 1 1 7 2\n"))
    np.testing.assert_array_equal(s.camera.position, [1, 1, 1])
    assert s.root_number_shadow_rays == 7.0 and s.max_recursions == 2.0


def test_unknown_keyword_raises_value_error(tmp_path):
    with pytest.raises(ValueError, match="Unknown object type: tri"):
        R.parse_scene_file(write_scene(tmp_path, MINI + "tri 0 0 0 1 1 1 2 2 2 1\n"))


def test_short_line_raises(tmp_path):
    with pytest.raises(ValueError):
        R.parse_scene_file(write_scene(tmp_path, "sph 0 0 0 1\n"))


def test_camera_basis_is_orthonormal():
    cam = scene("transparency_v7").camera
    f, r, u = cam.forward, cam.right, cam.up
    for v in (f, r, u):
        assert abs(np.linalg.norm(v) - 1) < 1e-15
    assert abs(f @ r) < 1e-15 and abs(f @ u) < 1e-15 and abs(r @ u) < 1e-15


def test_pack_layout_is_component_major(tmp_path):
    s = R.parse_scene_file(write_scene(tmp_path, MINI))
    p = packing.pack_scene(s)
    d = p.desc
    assert d.struct_size == C.sizeof(packing.RtSceneDesc) and d.abi_version == packing.RT_ABI_VERSION
    assert (d.n_surfaces, d.n_materials, d.n_lights, d.n_shadow, d.max_rec) == (3, 2, 2, 3, 4)
    S = 3
    types = [d.surf_type[i] for i in range(S)]
    assert types == [packing.RT_SURF_BOX, packing.RT_SURF_PLANE_RAW, packing.RT_SURF_SPHERE]
    sp = np.array([d.surf_param[i] for i in range(4 * S)]).reshape(4, S)
    np.testing.assert_array_equal(sp[:, 0], [1, 2, 3, 2])        # box centre + edge
    np.testing.assert_array_equal(sp[:, 1], [0, 2, 0, -1])       # plane RAW normal + offset
    np.testing.assert_array_equal(sp[:, 2], [0, 0, 0, 1.5])      # sphere centre + radius
    assert [d.surf_material[i] for i in range(S)] == [0, 1, 0]   # 1-based -> 0-based
    mp = np.array([d.mat_param[i] for i in range(11 * 2)]).reshape(11, 2)
    np.testing.assert_array_equal(mp[:, 1], [0, 1, 0, 1, 1, 1, 0.5, 0.5, 0.5, 20, 0.25])
    lp = np.array([d.light_param[i] for i in range(9 * 2)]).reshape(9, 2)
    np.testing.assert_array_equal(lp[:, 1], [1, 5, 0, 1, 1, 1, 1, 0, 2])
    assert list(d.background) == [0.1, 0.2, 0.3]


def test_material_index_semantics():
    # materials[material_index - 1]: 0 silently addresses the LAST material (Python's [-1]); kept
    assert packing._material_slot(1, 5) == 0
    assert packing._material_slot(5, 5) == 4
    assert packing._material_slot(0, 5) == 4
    assert packing._material_slot(-1, 5) == 3
    with pytest.raises(IndexError):
        packing._material_slot(6, 5)
    with pytest.raises(IndexError):
        packing._material_slot(-5, 5)


def test_effective_depth_and_shadow_count():
    assert packing.effective_depth(10.0) == 10
    assert packing.effective_depth(2.5) == 3      # depth >= 2.5 first true at 3
    assert packing.effective_depth(0) == 0 and packing.effective_depth(-3) == 0


def test_pack_rejects_incomplete_scenes(tmp_path):
    with pytest.raises(ValueError, match="camera"):
        packing.pack_scene(R.parse_scene_file(write_scene(tmp_path, "set 1 1 1 1 1\n")))
    with pytest.raises(ValueError, match="set"):
        packing.pack_scene(R.parse_scene_file(write_scene(tmp_path, "cam 0 0 5 0 0 0 0 1 0 1 1\n")))
    with pytest.raises(ValueError, match="root_number_shadow_rays"):
        packing.pack_scene(R.parse_scene_file(write_scene(tmp_path, "cam 0 0 5 0 0 0 0 1 0 1 1\nset 1 1 1 0 1\n")))


def test_row_selection_matches_stripe_formula():
    rows = packing.selected_rows(10, 40, 3, 37, stripe_rows=4, stripe_count=3, stripe_index=1)
    assert rows == [y for y in range(3, 37) if (y // 4) % 3 == 1]
    all_rows = sorted(sum((packing.selected_rows(10, 40, 0, 40, 4, 3, r) for r in range(3)), []))
    assert all_rows == list(range(40))


@pytest.mark.reference
def test_packer_accepts_the_references_own_objects():
    """Duck typing: a SceneSettings built by the reference's parser packs to the same numbers."""
    from oracle import ref_harness as H
    if not H.available():
        pytest.skip("reference tree not present")
    for name in ("pool", "cubes", "transparency_v7"):
        mine = packing.pack_scene(scene(name)).desc
        import os
        theirs = packing.pack_scene(H.parse(os.path.join(os.path.dirname(os.path.dirname(__file__)), "scenes", name + ".txt"))).desc
        S = mine.n_surfaces
        assert (mine.n_surfaces, mine.n_materials, mine.n_lights) == (theirs.n_surfaces, theirs.n_materials, theirs.n_lights)
        a = np.array([mine.surf_param[i] for i in range(4 * S)]); b = np.array([theirs.surf_param[i] for i in range(4 * S)])
        np.testing.assert_array_equal(a, b)   # shipped scenes have axis-aligned unit normals
        assert [mine.surf_material[i] for i in range(S)] == [theirs.surf_material[i] for i in range(S)]
