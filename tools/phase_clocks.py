#!/usr/bin/env python3
"""Wave clocks per phase of the level kernel (a library built with -DRT_PHASE_CLOCKS, named by
RT_HIP_LIBRARY): phase_clocks.py WIDTH HEIGHT [DEPTH].  Prints the share of each phase and, for
frames of few tiles, the clocks per wave."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k  # noqa: E402

w, h = int(sys.argv[1]), int(sys.argv[2])
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
os.environ["RT_HIP_PIPELINE"] = "fused"
ctx = RenderContext([0])
ctx.set_scene(scene_1k().flatten())
ctx.set_profiling(1)
buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
for _ in range(2):
    ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), 0)
    st = ctx.stats()
names = ["claim / wait", "record setup", "ball loops", "bitmaps", "sphere loops", "exact / plain shading", "planes",
         "results / push"]
keys = ["sphere_tests", "plane_tests", "triangle_tests", "exact_candidates", "filter_ray_tests", "filter_origin_terms",
        "primary_filter_ray_tests", "primary_exact_candidates"]
# (rt_get_stats adds the primary rays' analytic share to the first three: take it off)
prim = st["rays_primary"]
ns, npl, nt = 1000, 6, 0
clk = [st[k] for k in keys]
clk[0] -= prim * ns
clk[1] -= prim * npl
clk[2] -= prim * nt
total = sum(clk)
tiles_a = (st["rays_secondary"] + 255) // 256
print(f"{w}x{h} depth {depth}: {st['rays_secondary']} reflecting records (~{tiles_a} tiles), "
      f"{st['shaded_hits'] - st['rays_secondary']} plain records; trace {st['trace_ms'] * 1e3:.1f} us")
for n, c in zip(names, clk):
    print(f"  {n:24s} {100.0 * c / total:5.1f} %   {c / 100.0 / max(1, 4 * tiles_a):10.2f} us per wave of a reflecting tile"
          " (100 MHz clock)" if False else f"  {n:24s} {100.0 * c / total:5.1f} %   {c:14d} ticks")
ctx.close()
