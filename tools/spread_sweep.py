#!/usr/bin/env python3
"""Frame times of the 1k-sphere scene at a few frame sizes, one line (for sweeps of a knob such as
RT_HIP_SPREAD_BELOW set in the environment)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import Camera, RenderContext, scene_1k
ctx = RenderContext([0]); ctx.set_scene(scene_1k().flatten()); stream = torch.cuda.Stream()
out = []
for (w, h, depth) in ((3840, 2160, 4), (1920, 1080, 8), (1920, 1080, 4), (3840, 270, 4), (640, 360, 4)):
    buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
    for _ in range(3): ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(); n = 200; t0 = time.perf_counter()
    for _ in range(n): ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(); out.append(f"{w}x{h}d{depth} {(time.perf_counter()-t0)/n*1e3:.4f}")
print("  ".join(out))
