#!/usr/bin/env python3
"""Per-call times of rt_render (frame into host memory) and rt_render_rows_device on the 64k-sphere scene,
call by call: what the first calls after a change of entry point or stream cost."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import Camera, RenderContext, scene_64k
ctx = RenderContext([0]); ctx.set_scene(scene_64k().flatten())
W,H,D = 1920,1080,8
out = np.empty((H,W), dtype=np.int32)
buf = torch.empty((H, W), dtype=torch.int32, device="cuda"); st = torch.cuda.Stream()
for i in range(8):
    t=time.perf_counter(); ctx.render(Camera(), W,H,D, out=out); print("rt_render", round((time.perf_counter()-t)*1e3,3), flush=True)
for i in range(4):
    torch.cuda.synchronize(); t=time.perf_counter(); ctx.render_rows_device(Camera(), W, H, 0, H, D, buf.data_ptr(), st.cuda_stream); torch.cuda.synchronize(); print("rows_device", round((time.perf_counter()-t)*1e3,3), flush=True)
for i in range(3):
    t=time.perf_counter(); ctx.render(Camera(), W,H,D, out=out); print("rt_render", round((time.perf_counter()-t)*1e3,3), flush=True)
