#!/usr/bin/env python3
"""The 64k-sphere scene at several frame sizes (and one rank's interleaved share of the 1920x1080 frame), device-resident,
one frame at a time: c5_sizes.py   (RT_HIP_* knobs from the environment)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_64k  # noqa: E402

torch.cuda.set_device(0)
stream = torch.cuda.Stream()
ctx = RenderContext([0])
ctx.set_scene(scene_64k().flatten())
cam = Camera()
buf = torch.empty((1080, 1920), dtype=torch.int32, device="cuda")


def timed(call, n=30):
    for _ in range(5):
        call()
    ctx.frame_status()
    t = time.perf_counter()
    for _ in range(n):
        call()
    ctx.frame_status()
    return (time.perf_counter() - t) / n * 1e3


out = []
for w, h in ((1920, 1080), (1280, 720), (960, 540), (480, 270)):
    out.append(f"{w}x{h} {timed(lambda: ctx.render_rows_device(cam, w, h, 0, h, 8, buf.data_ptr(), stream.cuda_stream)):.3f}")
for parts in (2, 4, 8):
    out.append(f"1/{parts} {timed(lambda: ctx.render_bands_device(cam, 1920, 1080, 16, 0, parts, 8, buf.data_ptr(), stream.cuda_stream)):.3f}")
print("  ".join(out), flush=True)
ctx.close()
