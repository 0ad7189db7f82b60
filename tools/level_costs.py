#!/usr/bin/env python3
"""Light-ray work per level: shadow rays and exact occluder tests of the first levels of a benchmark
scene (rendered with the work counters on at depth 0, 1, 2, ...; the differences belong to the
level that was added).  level_costs.py [c3|c5] [max_depth]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k  # noqa: E402
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h, scene = (3840, 2160, scene_1k) if cfg == "c3" else (1920, 1080, scene_64k)
ctx = RenderContext([0])
ctx.set_scene(scene().flatten())
ctx.set_profiling(1)
buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
prev = None
for depth in range(top + 1):
    ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), 0)
    st = ctx.stats()
    cur = (st["rays_shadow"], st["shadow_candidates"], st["shaded_hits"], st["rays_secondary"])
    if prev is not None:
        rays, cand = cur[0] - prev[0], cur[1] - prev[1]
        print(f"level {depth}: {cur[2] - prev[2]} records, {rays} light rays, {cand} exact tests = {cand / max(1, rays):.1f} per ray")
    else:
        print(f"level 0: {cur[2]} records, {cur[0]} light rays, {cur[1]} exact tests = {cur[1] / max(1, cur[0]):.1f} per ray")
    prev = cur
ctx.close()
