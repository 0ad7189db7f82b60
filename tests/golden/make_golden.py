#!/usr/bin/env python3
"""Writes the golden fixtures in this directory from the CPU oracle (oracle/rt_oracle.c).

The reference (C#/.NET + OpenTK) cannot be built or run here and ships no vectors of its
own, so these files pin the ORACLE (and through it the GPU library) against regressions;
they are not outputs of the reference itself.  Run from the repository root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from raytracer_amd import Camera, default_scene, scene_1k  # noqa: E402
from tests.conftest import synthetic_texture  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    O.build()
    cam = Camera()
    frames = {}
    frames["default_cyan_96"], _ = O.render(default_scene().flatten(), cam, 96, 96, max_depth=5)
    frames["default_textured_96"], _ = O.render(default_scene(textured_wall=True).flatten(), cam, 96, 96,
                                                max_depth=5, textures=[synthetic_texture()])
    frames["default_cyan_160x90_depth0"], _ = O.render(default_scene().flatten(), cam, 160, 90, max_depth=0)
    frames["spheres1k_64x36_depth4"], cnt = O.render(scene_1k().flatten(), cam, 64, 36, max_depth=4)
    moved = Camera(position=(-1.0, 0.2, 1.0), direction=(0.6, 0.0, 0.8), fov=70.0)
    frames["default_moved_camera_80x60"], _ = O.render(default_scene().flatten(), moved, 80, 60, max_depth=5)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **frames)

    # Per-function vectors: ray directions and closest hits for a grid of pixels.
    fs = default_scene().flatten()
    rows = []
    for (w, h) in ((1080, 1080), (1920, 1080)):
        for y in range(0, h, h // 6):
            for x in range(0, w, w // 6):
                d = O.ray_direction(cam, x, y, w, h)
                prim, t, hit = O.intersect(fs, (0, 0, 0), d)
                rows.append([w, h, x, y, *d.view(np.uint32), prim, np.float32(t).view(np.uint32),
                             *hit.view(np.uint32)])
    np.save(os.path.join(HERE, "rays_default_scene.npy"), np.array(rows, dtype=np.int64))
    counters = {k: int(v) for k, v in cnt.items()}
    np.save(os.path.join(HERE, "spheres1k_64x36_depth4_counters.npy"),
            np.array([counters[k] for k in sorted(counters)], dtype=np.int64))
    print("wrote", sorted(frames), len(rows), "ray vectors")


if __name__ == "__main__":
    main()
