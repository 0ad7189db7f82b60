#!/usr/bin/env python3
"""Frame times of the 1k-sphere scene at several frame sizes and depths, one frame at a time:
small_frames.py  (RT_HIP_GRID=0 in the environment: without the spheres' grid)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k  # noqa: E402

for name, scene in (("1k", scene_1k), ("64k", scene_64k)):
    ctx = RenderContext([0])
    ctx.set_scene(scene().flatten())
    stream = torch.cuda.Stream()
    for (w, h, depth) in ((256, 256, 4), (640, 360, 4), (1280, 720, 4), (1920, 1080, 4), (1920, 1080, 8), (3840, 270, 4), (3840, 540, 4),
                          (3840, 1080, 4), (3840, 2160, 4)):
        if name == "64k" and w * h > 1920 * 1080:
            continue
        buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
        for _ in range(3):
            ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        print(f"{name:>4} {w}x{h} depth {depth}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms/frame", flush=True)
    ctx.close()
