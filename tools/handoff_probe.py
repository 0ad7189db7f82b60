#!/usr/bin/env python3
"""What a 3840x2160 frame costs until it lies in HOST memory, by hand-off form (one GPU).

  d2h_only      one copy of the finished frame, device -> page-locked host memory, alone
  resident      the frame rendered into device memory (no hand-off), one frame at a time, blocking
  zero_copy     the kernels store the pixels straight into page-locked host memory (the pointer of a
                pinned buffer handed to rt_render_rows_device as the frame)
  rt_render     whole rt_render calls into the library's page-locked framebuffer (in THIS process, i.e. on the
                PyTorch wheel's HIP runtime, whose copies are shader kernels; tools/handoff_host.py: without PyTorch)

Every figure is wall-clock per blocking call, the mean of REPS calls after warm-up calls.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k, default_scene  # noqa: E402

REPS = int(os.environ.get("PROBE_REPS", "30"))


def timed(fn, reps=REPS, warm=5):
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e3


def main() -> None:
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    W, H, D, scene = {"c3": (3840, 2160, 4, scene_1k), "c4": (7680, 4320, 4, scene_1k),
                      "c2": (1920, 1080, 0, default_scene), "c5": (1920, 1080, 8, scene_64k)}[which]
    torch.cuda.set_device(0)
    cam = Camera()
    ctx = RenderContext([0])
    ctx.set_scene(scene().flatten())
    dev = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    host = torch.zeros((H, W), dtype=torch.int32).pin_memory()
    stream = torch.cuda.Stream()

    # warm the clocks
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        ctx.render_rows_device(cam, W, H, 0, H, D, dev.data_ptr(), stream.cuda_stream)
        ctx.frame_status()

    def resident():
        ctx.render_rows_device(cam, W, H, 0, H, D, dev.data_ptr(), stream.cuda_stream)
        ctx.frame_status()

    def d2h():
        with torch.cuda.stream(stream):
            host.copy_(dev, non_blocking=True)
        stream.synchronize()

    def zero_copy():
        ctx.render_rows_device(cam, W, H, 0, H, D, host.data_ptr(), stream.cuda_stream)
        ctx.frame_status()

    out = {"config": which, "frame": f"{W}x{H}", "bytes": W * H * 4}
    out["resident_ms"] = round(timed(resident), 4)
    out["d2h_only_ms"] = round(timed(d2h), 4)
    out["d2h_GBps"] = round(W * H * 4 / (out["d2h_only_ms"] * 1e-3) / 1e9, 2)
    resident()
    ref = dev.cpu().numpy().copy()
    host.zero_()
    out["zero_copy_ms"] = round(timed(zero_copy), 4)
    out["zero_copy_identical"] = bool((host.numpy() == ref).all())

    print(out, flush=True)
    owned = ctx.alloc_pixels(H, W)
    out = {"rt_render_library_framebuffer_ms": round(timed(lambda: ctx.render(cam, W, H, D, out=owned)), 4),
           "identical": bool((owned == ref).all())}
    print(out, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
