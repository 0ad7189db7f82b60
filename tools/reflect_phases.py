#!/usr/bin/env python3
"""Where a wave of k_reflect spends its time, per level (a library built with -DRT_PHASE_CLOCKS, named
by RT_HIP_LIBRARY; RT_HIP_LEVEL_KERNELS=1 so that every launch has the device to itself):
reflect_phases.py [c3|c5] [max_depth].  The frame is rendered at depth 1, 2, ...; the clocks added by
a depth belong to the level that was added.  Prints, per level, the mean time of a wave per phase."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w, h, scene = (3840, 2160, scene_1k) if cfg == "c3" else (1920, 1080, scene_64k)
ctx = RenderContext([0])
ctx.set_scene(scene().flatten())
ctx.set_profiling(2)
buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
names = ["claim / record load", "record setup", "ball loops", "bitmaps", "sphere loops", "exact tests", "planes",
         "results / push"]
keys = ["sphere_tests", "plane_tests", "triangle_tests", "exact_candidates", "filter_ray_tests", "filter_origin_terms",
        "primary_filter_ray_tests", "primary_exact_candidates"]
ns, npl = (1000, 6) if cfg == "c3" else (65536, 6)
prev, prev_sec = [0] * 8, 0
for depth in range(1, top + 1):
    for _ in range(2):
        ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), 0)
        st = ctx.stats()
    prim = st["rays_primary"]
    clk = [st[k] for k in keys]
    clk[0] -= prim * ns    # (rt_get_stats adds the primary rays' analytic share to the first three)
    clk[1] -= prim * npl
    sec = st["rays_secondary"]
    recs = sec - prev_sec
    waves = 4 * ((recs + 255) // 256)
    d = [a - b for a, b in zip(clk, prev)]
    print(f"level {depth - 1}: {recs} reflecting records, {waves} waves; per wave {sum(d) / 100.0 / waves:8.2f} us in all")
    for n, c in zip(names, d):
        print(f"    {n:22s} {c / 100.0 / waves:8.2f} us  {100.0 * c / max(1, sum(d)):5.1f} %")
    prev, prev_sec = clk, sec
ctx.close()
