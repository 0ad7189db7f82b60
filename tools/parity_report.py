#!/usr/bin/env python3
"""Parity report: the library (through the C ABI) against the CPU restatement on the five
benchmark configurations, with the mismatch map of each.

    python tools/parity_report.py [--out gpurun_out/parity] [--quick]

Per configuration: share of pixels that are identical, share within +-1 per 8-bit channel (the
acceptance bar is 99.9 %), the largest channel difference, the histogram of differences, and
whether the ray / Intersect-call counters agree.  The mismatch map (per-pixel maximum channel
difference, 0 = identical) is written as a binary PGM next to the JSON summary.  Needs a GPU.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402  (the checker; this is test infrastructure)
from raytracer_amd import Camera, RenderContext, default_scene, scene_1k, scene_64k  # noqa: E402
from tests.parity import channel_diff  # noqa: E402


def write_pgm(path: str, diff: np.ndarray) -> None:
    img = np.minimum(diff, 255).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity"))
    ap.add_argument("--quick", action="store_true", help="smaller frames (seconds instead of minutes)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    O.build()
    q = args.quick
    cases = [
        ("C1", "reference default scene, template resolution, depth 5", default_scene(), (1080, 1080), 5, None),
        ("C2", "default scene 1920x1080, primary + shadow rays only", default_scene(), (1920, 1080), 0, None),
        ("C3", "1k spheres 3840x2160, 4 bounces", scene_1k(), (960, 540) if q else (3840, 2160), 4, None),
        ("C4", "1k spheres 7680x4320, 4 bounces, eight blocks of 540 rows", scene_1k(),
         (1920, 1080) if q else (7680, 4320), 4, 8),
        ("C5", "64k spheres, 16 lights, 8 bounces", scene_64k(), (96, 54) if q else (320, 180), 8, None),
    ]
    cam = Camera()
    report = []
    with RenderContext([0]) as ctx:
        for name, what, scene, (w, h), depth, blocks in cases:
            flat = scene.flatten()
            ctx.set_scene(flat)
            ctx.set_profiling(True)
            if blocks:
                import torch
                frame = torch.empty((h, w), dtype=torch.int32, device="cuda")
                for r in range(blocks):
                    y0, y1 = h * r // blocks, h * (r + 1) // blocks
                    ctx.render_rows_device(cam, w, h, y0, y1, depth, frame[y0:y1].data_ptr(), 0)
                    torch.cuda.synchronize()
                gpu, st = frame.cpu().numpy(), None
            else:
                gpu = ctx.render(cam, w, h, depth)
                st = ctx.stats()
                # the same frame the way the host renders it (no counters: light rays of
                # multi-chunk scenes stop at any occluder) must be the same frame
                ctx.set_profiling(False)
                unprofiled_same = bool(np.array_equal(ctx.render(cam, w, h, depth), gpu))
                ctx.set_profiling(True)
            t0 = time.perf_counter()
            ref, cnt = O.render(flat, cam, w, h, max_depth=depth)
            cpu_s = time.perf_counter() - t0
            d = channel_diff(gpu, ref)
            hist = np.bincount(np.minimum(d, 255).ravel(), minlength=4)
            entry = {
                "config": name, "what": what, "size": f"{w}x{h}", "max_depth": depth, "pixels": int(d.size),
                "identical": float((gpu == ref).mean()), "within_1": float((d <= 1).mean()),
                "max_channel_diff": int(d.max()),
                "diff_histogram": {str(k): int(v) for k, v in enumerate(hist) if v},
                "mismatch_map": f"{name}_mismatch.pgm", "cpu_restatement_seconds": round(cpu_s, 2),
            }
            if st is not None:
                keys = ("rays_primary", "rays_shadow", "rays_secondary", "shaded_hits", "sphere_tests",
                        "plane_tests", "triangle_tests")
                entry["counters_agree"] = all(int(st[k]) == int(cnt[k]) for k in keys)
                entry["rays"] = int(st["rays_primary"] + st["rays_shadow"] + st["rays_secondary"])
                entry["identical_without_profiling"] = unprofiled_same
            write_pgm(os.path.join(args.out, entry["mismatch_map"]), d)
            report.append(entry)
            print(json.dumps(entry), flush=True)
    with open(os.path.join(args.out, "parity_report.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
