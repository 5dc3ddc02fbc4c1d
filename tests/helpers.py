"""Shared helpers of the test-suite."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(ROOT, "scenes")
THREADS = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))


def scene(name):
    import raytracer_amd as R
    return R.parse_scene_file(os.path.join(SCENES, name + ".txt"))


def golden_png(rel):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLD, rel)))


def golden_npz(rel):
    return np.load(os.path.join(GOLD, rel), allow_pickle=False)


def golden_json(rel):
    with open(os.path.join(GOLD, rel)) as f:
        return json.load(f)


def grid(w, h):
    return np.tile(np.arange(w, dtype=np.int32), h), np.repeat(np.arange(h, dtype=np.int32), w)


def write_scene(tmp_path, text, name="scene.txt"):
    p = os.path.join(str(tmp_path), name)
    with open(p, "w") as f:
        f.write(text)
    return p
