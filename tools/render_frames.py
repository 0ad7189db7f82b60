#!/usr/bin/env python3
"""A few blocking rt_render calls into the library's page-locked framebuffer, 2 ms apart (for profilers):
render_frames.py [c3|c4|c2|c5] [frames]."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k, default_scene  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w, h, depth, scene = {"c3": (3840, 2160, 4, scene_1k), "c4": (7680, 4320, 4, scene_1k),
                      "c2": (1920, 1080, 0, default_scene), "c5": (1920, 1080, 8, scene_64k)}[cfg]
ctx = RenderContext([0])
ctx.set_scene(scene().flatten())
out = ctx.alloc_pixels(h, w)
for i in range(frames):
    t = time.perf_counter()
    ctx.render(Camera(), w, h, depth, out=out)
    print(f"frame {i}: {(time.perf_counter() - t) * 1e3:.3f} ms", flush=True)
    time.sleep(0.002)
ctx.close()
