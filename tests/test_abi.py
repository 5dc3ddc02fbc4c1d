"""The C ABI: every function declared in include/rt_amd.h is exported, structs match their ctypes mirrors."""
import ctypes as C
import os
import re
import subprocess

import pytest

from helpers import ROOT
from raytracer_amd import packing, runtime

HEADER = os.path.join(ROOT, "include", "rt_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    assert set(declared_functions()) == set(runtime.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = runtime.load_library()  # raises if the .so has not been built
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "rt_amd.h"\n'
        "int main(void){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(rt_scene_desc), sizeof(rt_render_params),"
        " sizeof(rt_stats), offsetof(rt_scene_desc, surf_type), offsetof(rt_scene_desc, n_shadow),"
        " offsetof(rt_stats, ms_total), offsetof(rt_render_params, seed), offsetof(rt_scene_desc, light_param));return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(packing.RtSceneDesc), C.sizeof(packing.RtRenderParams), C.sizeof(packing.RtStats),
            packing.RtSceneDesc.surf_type.offset, packing.RtSceneDesc.n_shadow.offset, packing.RtStats.ms_total.offset,
            packing.RtRenderParams.seed.offset, packing.RtSceneDesc.light_param.offset]
    assert got == want


def test_rows_selected_is_a_pure_host_function():
    lib = runtime.load_library()
    p = packing.render_params(10, 40, 3, 37, 4, 3, 1)
    assert lib.rt_rows_selected(C.byref(p)) == len(packing.selected_rows(10, 40, 3, 37, 4, 3, 1))


def test_product_never_imports_the_oracle():
    """The shipped package must not reach into oracle/ (checker only)."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "rt_oracle" not in text, f


def test_no_gpu_means_a_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(runtime.RtError):
        runtime.Renderer(0)
    import raytracer_amd as R
    from helpers import scene
    with pytest.raises(runtime.RtError):
        R.render(scene("pool"), 8, 8, verbose=False)
