#!/usr/bin/env python3
"""Does where the scene's tables and the ray queues land in device memory change a frame's time?

placement_probe.py <order>   order = fresh | after8k | after8k_newctx | after8k_shrunk
  fresh            the 64k-sphere, 16-light scene at 1920x1080 in a fresh process
  after8k          the same context has rendered the 1k-sphere scene at 7680x4320 before (its ray queues stay)
  after8k_newctx   that context was destroyed first; a new one renders the 64k-sphere scene
Prints ms per frame (device-resident, one frame at a time) and, with RT_HIP_SCENE_TRACE=1, the library's own report.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k  # noqa: E402


def frames(ctx, w, h, depth, n, buf, stream):
    cam = Camera()
    for _ in range(5):
        ctx.render_rows_device(cam, w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
    ctx.frame_status()
    t = time.perf_counter()
    for _ in range(n):
        ctx.render_rows_device(cam, w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
    ctx.frame_status()
    return (time.perf_counter() - t) / n * 1e3


def main():
    order = sys.argv[1] if len(sys.argv) > 1 else "fresh"
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    ctx = RenderContext([0])
    if order != "fresh":
        big = torch.empty((4320, 7680), dtype=torch.int32, device="cuda")
        ctx.set_scene(scene_1k().flatten())
        print(f"8K frame: {frames(ctx, 7680, 4320, 4, 20, big, stream):.3f} ms", flush=True)
        del big
        if order == "after8k_newctx":
            ctx.close()
            torch.cuda.empty_cache()
            ctx = RenderContext([0])
    buf = torch.empty((1080, 1920), dtype=torch.int32, device="cuda")
    ctx.set_scene(scene_64k().flatten())
    for rep in range(3):
        print(f"{order}: 64k-sphere scene 1920x1080 {frames(ctx, 1920, 1080, 8, 40, buf, stream):.3f} ms", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
