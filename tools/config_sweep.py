#!/usr/bin/env python3
"""Frame times of the benchmark configurations on one GPU, and what a row partition over
G GPUs would cost per rank (each rank's rows rendered alone on this GPU, one after the other).

    python tools/config_sweep.py [--configs c2,c3,c5,split] [--frames 5]

Prints one JSON object per measurement.  Device-resident timing (rt_render_rows_device on
torch's current stream, wall clock around a synchronised batch of frames).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracer_amd import Camera, RenderContext, default_scene, scene_1k, scene_64k  # noqa: E402


def time_rows(ctx, cam, w, h, depth, rows, frames, profile=True, bands=None):
    """ms per render call, each part rendered alone, back to back: `rows` = [(y0, y1), ...]
    contiguous blocks, or with bands = (band_rows, parts) the parts of the interleaved partition."""
    if bands:
        band, parts = bands
        jobs = [(lambda buf, s, p=p: ctx.render_bands_device(cam, w, h, band, p, parts, depth, buf.data_ptr(), s),
                 (h + parts - 1) // parts + band) for p in range(parts)]
    else:
        jobs = [(lambda buf, s, y0=y0, y1=y1: ctx.render_rows_device(cam, w, h, y0, y1, depth, buf.data_ptr(), s),
                 y1 - y0) for y0, y1 in rows]
    stream = torch.cuda.Stream()
    out = []
    for call, nrows in jobs:
        buf = torch.empty((nrows, w), dtype=torch.int32, device="cuda")
        ctx.set_profiling(False)
        for _ in range(3):   # (buffers, streams and the scene's tables settle during the first calls)
            call(buf, stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            call(buf, stream.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / frames * 1e3
        st = None
        if profile:
            ctx.set_profiling(True)
            call(buf, stream.cuda_stream)
            st = ctx.stats()
        out.append((ms, st))
    return out


def rays_of(st):
    return st["rays_primary"] + st["rays_shadow"] + st["rays_secondary"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c2,c3,c5,split")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--c5-size", default="1920x1080")
    ap.add_argument("--split-size", default="3840x2160", help="frame of the simulated row partitions")
    ap.add_argument("--host-size", default="3840x2160", help="frame of the rt_render-into-host-memory measurement")
    args = ap.parse_args()
    want = set(args.configs.split(","))
    cam = Camera()
    ctx = RenderContext([0])

    def single(name, flat, w, h, depth, frames):
        ctx.set_scene(flat)
        (ms, st), = time_rows(ctx, cam, w, h, depth, [(0, h)], frames)
        print(json.dumps({"config": name, "size": f"{w}x{h}", "max_depth": depth, "ms_per_frame": round(ms, 4),
                          "Mrays_per_s": round(rays_of(st) / ms / 1e3, 1), "rays": rays_of(st),
                          "primary_ms": round(st["primary_ms"], 4), "trace_ms": round(st["trace_ms"], 4),
                          "shade_ms": round(st["shade_ms"], 4),
                          "filter_ray_tests": st["filter_ray_tests"], "sphere_tests": st["sphere_tests"],
                          "exact_candidates": st["exact_candidates"]}), flush=True)

    if "c2" in want:
        single("C2 default scene, primary + shadow only", default_scene().flatten(), 1920, 1080, 0, args.frames * 4)
    if "c3" in want:
        single("C3 1k spheres, 4 bounces", scene_1k().flatten(), 3840, 2160, 4, args.frames * 4)
    if "c4" in want:
        single("C4 resolution on one GPU", scene_1k().flatten(), 7680, 4320, 4, args.frames)
    if "c5" in want:
        w, h = (int(v) for v in args.c5_size.split("x"))
        single("C5 64k spheres, 16 lights, 8 bounces", scene_64k().flatten(), w, h, 8, max(1, args.frames // 3))
    if "host" in want:
        # rt_render into host memory (what the C# host calls): pageable array vs page-locked array.
        import numpy as np
        ctx.set_scene(scene_1k().flatten())
        ctx.set_profiling(False)
        (w, h), depth = (int(v) for v in args.host_size.split("x")), 4
        out = np.zeros(w * h, dtype=np.int32)
        for pinned in (False, True):
            if pinned:
                try:
                    ctx.pin_pixels(out)
                except Exception as e:   # systems that do not page-lock user memory
                    print(json.dumps({"config": f"{w}x{h} through rt_render into host memory",
                                      "framebuffer": "page-locked (rt_pin_pixels)", "refused": str(e)}), flush=True)
                    break
            ctx.render(cam, w, h, depth, out=out)
            t0 = time.perf_counter()
            for _ in range(args.frames * 2):
                ctx.render(cam, w, h, depth, out=out)
            ms = (time.perf_counter() - t0) / (args.frames * 2) * 1e3
            ctx.set_profiling(True)
            ctx.render(cam, w, h, depth, out=out)
            st = ctx.stats()
            ctx.set_profiling(False)
            print(json.dumps({"config": f"{w}x{h} through rt_render into host memory",
                              "framebuffer": "page-locked (rt_pin_pixels)" if pinned else "pageable",
                              "ms_per_frame": round(ms, 3), "kernel_ms": round(st["kernel_ms"], 3),
                              "copy_ms": round(st["copy_ms"], 3)}), flush=True)
        ctx.unpin_pixels()
    if "split" in want:
        ctx.set_scene(scene_1k().flatten())
        (w, h), depth = (int(v) for v in args.split_size.split("x")), 4
        (full_ms, _), = time_rows(ctx, cam, w, h, depth, [(0, h)], args.frames * 4, profile=False)
        for G, banded in ((2, False), (4, False), (8, False), (2, True), (4, True), (8, True)):
            rows = [((h * r) // G, (h * (r + 1)) // G) for r in range(G)]
            res = time_rows(ctx, cam, w, h, depth, rows, args.frames * 4, bands=(16, G) if banded else None)
            ms = [round(m, 4) for m, _ in res]
            what = f"{G} interleaved parts (bands of 16 rows)" if banded else f"{G} contiguous row blocks"
            print(json.dumps({"config": f"{w}x{h} frame as {what}", "full_frame_ms": round(full_ms, 4),
                              "block_ms": ms, "max_ms": max(ms), "sum_ms": round(sum(ms), 4),
                              "primary_ms": [round(st["primary_ms"], 3) for _, st in res],
                              "trace_ms": [round(st["trace_ms"], 3) for _, st in res],
                              "shade_ms": [round(st["shade_ms"], 3) for _, st in res],
                              "rays": [rays_of(st) for _, st in res],
                              "efficiency_if_no_gather": round(full_ms / (G * max(ms)), 4)}), flush=True)
    if "pipe" in want:
        # Frames in flight: part 0 of G interleaved parts of the frame (what one rank of G renders),
        # rendered back to back by 1 .. 8 contexts taking turns, each on a stream of its own and
        # sized for its share of the device.  ms per frame = wall time / frames.
        flat = scene_1k().flatten()
        (w, h), depth = (int(v) for v in args.split_size.split("x")), 4
        for G in (8, 4, 2, 1):
            rows = (h + G - 1) // G + 16
            for n_ctx, share in ((1, 1), (2, 2), (3, 3), (4, 4), (6, 6), (8, 8), (6, 4), (8, 4)):
                ctxs = [RenderContext([0]) for _ in range(n_ctx)]
                streams = [torch.cuda.Stream() for _ in range(n_ctx)]
                bufs = [torch.empty((rows, w), dtype=torch.int32, device="cuda") for _ in range(n_ctx)]
                for c in ctxs:
                    c.set_scene(flat)
                    c.set_device_share(share)
                n = args.frames * 8

                def run(count):
                    for i in range(count):
                        k = i % n_ctx
                        ctxs[k].render_bands_device(cam, w, h, 16, 0, G, depth, bufs[k].data_ptr(), streams[k].cuda_stream)
                    torch.cuda.synchronize()

                run(2 * n_ctx)
                t0 = time.perf_counter()
                run(n)
                ms = (time.perf_counter() - t0) / n * 1e3
                for c in ctxs:
                    c.frame_status()
                    c.close()
                print(json.dumps({"config": f"{w}x{h}, part 0 of {G} interleaved parts, {n_ctx} contexts in turn, "
                                            f"device share {share}", "ms_per_frame": round(ms, 4)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
