#!/usr/bin/env python3
"""Renders a few frames of a benchmark configuration (for profilers): one_frame.py [c3|c5] [frames]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h, depth, scene = (3840, 2160, 4, scene_1k) if cfg == "c3" else (1920, 1080, 8, scene_64k)
ctx = RenderContext([0])
ctx.set_scene(scene().flatten())
buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
stream = torch.cuda.Stream()
for _ in range(frames):
    ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
    ctx.frame_status()
ctx.close()
