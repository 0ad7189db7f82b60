#!/usr/bin/env python3
"""Time of rt_set_scene (packing, the lights' tables, the grid, the uploads): set_scene_ms.py [c3|c5]."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderContext, scene_1k, scene_64k  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
fs = (scene_1k if cfg == "c3" else scene_64k)().flatten()
with RenderContext([0]) as ctx:
    for _ in range(2):
        t0 = time.perf_counter()
        ctx.set_scene(fs)
        print(f"{cfg}: rt_set_scene {1e3 * (time.perf_counter() - t0):.1f} ms")
