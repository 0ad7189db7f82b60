#!/usr/bin/env python3
"""Records per level of a benchmark configuration's frame, from the Intersect-call counters of frames
rendered at growing depth limits: level_counts.py [c3|c5]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
w, h, depth, scene = (3840, 2160, 4, scene_1k) if cfg == "c3" else (1920, 1080, 8, scene_64k)
with RenderContext([0]) as ctx:
    ctx.set_profiling(True)
    ctx.set_scene(scene().flatten())
    prev = None
    for d in range(depth + 1):
        ctx.render(Camera(), w, h, d)
        st = ctx.stats()
        cur = {k: st[k] for k in ("shaded_hits", "rays_secondary", "rays_shadow", "sphere_tests")}
        if prev is None:
            print(f"level 0: {cur['shaded_hits']} records shaded, {cur['rays_shadow']} shadow rays")
        else:
            print(f"level {d}: {cur['shaded_hits'] - prev['shaded_hits']} records shaded, "
                  f"{cur['rays_secondary'] - prev['rays_secondary']} reflection rays traced into it, "
                  f"{cur['rays_shadow'] - prev['rays_shadow']} shadow rays")
        prev = cur
