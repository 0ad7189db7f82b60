#!/usr/bin/env python3
"""Frame times of 1000 spheres under 3, 4 and 8 lights (3840x2160 depth 4, 1920x1080 depth 8), for knob comparisons."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, synthetic_scene, Light
for nl in (3, 4, 8):
    sc = synthetic_scene(1000, seed=5)
    sc.lights.clear()
    for i in range(nl):
        sc.Add(Light(-2.0 + 1.2 * i, -1.0, 0.5 + 0.3 * i))
    fs = sc.flatten()
    for (w, h, depth) in ((3840, 2160, 4), (1920, 1080, 8)):
        buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
        stream = torch.cuda.Stream()
        with RenderContext([0]) as ctx:
            ctx.set_scene(fs)
            for _ in range(5):
                ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); reps = 40
            for _ in range(reps):
                ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            print(f"{nl} lights {w}x{h} depth {depth}: {1e3 * (time.perf_counter() - t0) / reps:.4f} ms/frame", flush=True)
