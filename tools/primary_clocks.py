#!/usr/bin/env python3
"""Wave clocks per phase of the primary kernel (library built with -DRT_PRIMARY_CLOCKS, named by
RT_HIP_LIBRARY): primary_clocks.py [c3|c5] [WxH]."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k  # noqa: E402
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
w, h, depth, scene = (3840, 2160, 4, scene_1k) if cfg == "c3" else (1920, 1080, 8, scene_64k)
if len(sys.argv) > 2:
    w, h = (int(v) for v in sys.argv[2].split("x"))
flat = scene().flatten()
ns = int((flat.prim_types == 0).sum()); npl = int((flat.prim_types == 1).sum()); nt = int((flat.prim_types == 2).sum())
ctx = RenderContext([0])
ctx.set_scene(flat)
ctx.set_profiling(2)
buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
for _ in range(3):
    ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), 0)
    st = ctx.stats()
prim = st["rays_primary"]
names = ["claim + directions", "cone", "ball tests", "sphere loads + tests", "exact", "planes / triangles", "push"]
clk = [st["sphere_tests"] - prim * ns, st["plane_tests"] - prim * npl, st["triangle_tests"] - prim * nt, st["exact_candidates"],
       st["filter_ray_tests"], st["filter_origin_terms"], st["primary_filter_ray_tests"]]
tot = sum(clk)
tiles = ((w + 15) // 16) * ((h + 15) // 16)
print(f"{cfg} {w}x{h}: primary {st['primary_ms'] * 1e3:.1f} us; {tiles} wave tiles; exact calls per lane-tile "
      f"{st['primary_exact_candidates'] / (tiles * 64):.2f}; steps per tile {st['shadow_candidates'] / 64 / tiles:.2f}")
for n, c in zip(names, clk):
    print(f"  {n:24s} {100.0 * c / tot:5.1f} %  {c / tiles:10.1f} ticks per tile")
ctx.close()
