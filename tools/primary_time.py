#!/usr/bin/env python3
"""HIP-event time of the primary kernel alone (profiling mode 2) on a benchmark configuration."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k  # noqa: E402
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
w, h, depth, scene = (3840, 2160, 4, scene_1k) if cfg == "c3" else (1920, 1080, 8, scene_64k)
if len(sys.argv) > 2:
    w, h = (int(v) for v in sys.argv[2].split("x"))
ctx = RenderContext([0])
ctx.set_scene(scene().flatten())
ctx.set_profiling(2)
buf = torch.empty((h, w), dtype=torch.int32, device="cuda")
stream = torch.cuda.Stream()
acc = []
for i in range(8):
    ctx.render_rows_device(Camera(), w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
    st = ctx.stats()
    if i >= 3:
        acc.append((st["primary_ms"], st["trace_ms"], st["shade_ms"], st["kernel_ms"]))
print(cfg, w, h, "primary/trace/shade/all ms:", [round(sum(c) / len(c), 4) for c in zip(*acc)])
ctx.close()
