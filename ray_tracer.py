#!/usr/bin/env python3
"""Command line of the reference (`python ray_tracer.py scene.txt out.png --width W --height H`), served by the MI355X path.

The implementation lives in raytracer_amd/ray_tracer.py (parse_scene_file, render, main: counterparts of
/root/reference/ray_tracer.py:371-419, :297-368, :427-440).
"""
from raytracer_amd.ray_tracer import main, parse_scene_file, render  # noqa: F401

if __name__ == "__main__":
    main()
