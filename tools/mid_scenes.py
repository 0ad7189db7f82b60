#!/usr/bin/env python3
"""Frame times of scenes between the two benchmark configurations (synthetic scenes of N spheres, 16 lights,
1920x1080, 6 bounces), for knob comparisons: mid_scenes.py [N ...]."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, synthetic_scene  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 16384]
w, h, depth = 1920, 1080, 6
buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
stream = torch.cuda.Stream()
for n in sizes:
    radius = (0.02, 0.08) if n <= 4096 else (0.01, 0.04)
    fs = synthetic_scene(n, radius=radius, seed=91, lights="grid16").flatten()
    with RenderContext([0]) as ctx:
        t0 = time.perf_counter()
        ctx.set_scene(fs)
        t_scene = time.perf_counter() - t0
        for _ in range(5):
            ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
        ctx.frame_status()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 30
        for _ in range(reps):
            ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
            ctx.frame_status()
        torch.cuda.synchronize()
        print(f"{n:6d} spheres: {1e3 * (time.perf_counter() - t0) / reps:.4f} ms/frame, rt_set_scene {1e3 * t_scene:.0f} ms", flush=True)
