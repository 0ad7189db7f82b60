#!/usr/bin/env python3
"""bench.py — Mrays/s of the render loop on the 1k-sphere scene at 3840x2160, 4 bounces.

One process per GPU (`python bench.py` for one GPU; the driver launches N ranks through
torch.distributed.run).  A step is one frame: every rank renders its rows of the SAME
3840x2160 frame (every N-th band of 16 rows: rt_render_bands_device; scene and scratch
resident in HBM), then the blocks are gathered on rank 0 with one RCCL collective
(torch.distributed backend "nccl") and put back in row order.  Total work is fixed as N
grows: strong scaling.  The timed frames run without the library's counters; the ray counts
and per-kernel HIP-event times come from one more frame rendered afterwards with them on.

Printed by rank 0: one JSON line with
  value        whole-job Mrays/s = (primary + shadow + used reflection rays per frame,
               de-duplicated as include/rt_hip.h defines them) / seconds per frame
  roofline     the ray-traversal kernel (k_trace_levels, every level): algorithmic FLOP of
               the Intersect calls the reference algorithm makes (25 per ray-sphere test,
               13 per ray-plane, 51 per ray-triangle) / its HIP-event time in the last
               timed frame, against the 157.3 TFLOP/s vector-FP32 peak.  The kernels skip
               groups of spheres whose bounding ball no ray of a wave can meet, so the
               algorithmic rate can exceed the peak; "executed" gives the FLOP of the filter
               evaluations and exact candidates the kernel really ran over the same time.
  cpu_baseline the CPU restatement (oracle/, literal per-light recursion like the
               reference, all host threads) timed on the same frame
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (imported before librt_hip.so so both share one HIP runtime)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytracer_amd import Camera, RenderContext, scene_1k  # noqa: E402
from raytracer_amd.distributed import FramePipeline, render_block  # noqa: E402

WIDTH, HEIGHT, MAX_DEPTH = 3840, 2160, 4
PEAK_FP32_TFLOPS = 157.3
FLOP_SPHERE, FLOP_PLANE, FLOP_TRIANGLE, FLOP_LIGHT_SHADE = 25.0, 13.0, 51.0, 80.0


def cpu_baseline(flat, cam, row_step: int = 1) -> dict:
    """CPU path next to the GPU number: literal mode (reflection re-traced per light, as
    objects.cs:80-88 does), rows spread over the host threads like Parallel.For."""
    from oracle import oracle as O  # checker / baseline only; never on the product path
    O.build()
    threads = O.num_threads()
    t0 = time.perf_counter()
    O.render(flat, cam, WIDTH, HEIGHT, max_depth=MAX_DEPTH, mode=O.MODE_LITERAL, row_step=row_step)
    dt = time.perf_counter() - t0
    # Ray count of the same rows, de-duplicated like the GPU's (memoised mode, not timed).
    _, cnt = O.render(flat, cam, WIDTH, HEIGHT, max_depth=MAX_DEPTH, mode=O.MODE_MEMOISED,
                      row_step=row_step)
    rays = cnt["rays_primary"] + cnt["rays_shadow"] + cnt["rays_secondary"]
    return {
        "value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"{len(range(0, HEIGHT, row_step))} of the {HEIGHT} rows (every "
                  f"{row_step}) of the {WIDTH}x{HEIGHT} frame, literal per-light recursion, {dt:.1f} s",
    }


def pmc_traffic() -> tuple:
    """HBM bytes per launch of the traversal kernel on this workload, from the committed
    rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot be collected from inside this
    process); (None, reason) when the summary is missing."""
    path = os.path.join(ROOT, "profiles", "r01_v16_trace_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return int(d["traffic_bytes_per_launch"]), "profiles/r01_v16_trace_traffic.json (rocprofv3 --pmc, one GPU)"
    except (OSError, KeyError, ValueError):
        return None, "no PMC summary committed"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        dist = None

    cam = Camera()
    flat = scene_1k().flatten()
    ctx = RenderContext([local_rank])
    ctx.set_scene(flat)
    # Several ranks: two frames in flight, so that a frame's gather (and rank 0's row
    # permutation) runs while the next frame renders.
    pipeline = FramePipeline(WIDTH, HEIGHT, rank, world, torch.device("cuda", local_rank),
                             depth=2 if world > 1 else 1)

    def step() -> None:
        pipeline.render_frame(ctx, cam, MAX_DEPTH)

    def fence() -> None:
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0

    # Kernel durations of the SAME kernels the timed loop ran: a few more frames bracketed by HIP
    # events on the stream (profiling mode 2), averaged.
    ctx.set_profiling(2)
    ev = {"trace_ms": 0.0, "kernel_ms": 0.0, "primary_ms": 0.0, "shade_ms": 0.0}
    EVENT_FRAMES = 5
    for _ in range(EVENT_FRAMES):
        render_block(ctx, cam, WIDTH, HEIGHT, MAX_DEPTH, pipeline.begin())
        timed = ctx.stats()  # synchronises
        for k in ev:
            ev[k] += timed[k] / EVENT_FRAMES
    # One more frame with the device-side work counters on (profiling mode 1: the counting
    # instances of the kernels, ~2 % slower; only their counts are used).
    ctx.set_profiling(1)
    render_block(ctx, cam, WIDTH, HEIGHT, MAX_DEPTH, pipeline.begin())
    st = ctx.stats()
    keys = ("rays_primary", "rays_shadow", "rays_secondary", "shaded_hits", "sphere_tests",
            "plane_tests", "triangle_tests", "filter_ray_tests", "filter_origin_terms", "exact_candidates",
            "primary_filter_ray_tests", "primary_exact_candidates")
    totals = torch.tensor([float(st[k]) for k in keys], dtype=torch.float64, device="cuda")
    times = torch.tensor([elapsed, ev["trace_ms"], ev["kernel_ms"], ev["primary_ms"], ev["shade_ms"]],
                         dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    tot = dict(zip(keys, totals.tolist()))
    elapsed, trace_ms, kernel_ms, primary_ms, shade_ms = times.tolist()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays = tot["rays_primary"] + tot["rays_shadow"] + tot["rays_secondary"]
        ns, npl, nt = (int((flat.prim_types == t).sum()) for t in (0, 1, 2))
        # Tests made by the traversal kernels = all tests minus the primary rays' share.
        tr_s = tot["sphere_tests"] - tot["rays_primary"] * ns
        tr_p = tot["plane_tests"] - tot["rays_primary"] * npl
        tr_t = tot["triangle_tests"] - tot["rays_primary"] * nt
        trace_flop = FLOP_SPHERE * tr_s + FLOP_PLANE * tr_p + FLOP_TRIANGLE * tr_t
        frame_flop = (FLOP_SPHERE * tot["sphere_tests"] + FLOP_PLANE * tot["plane_tests"] +
                      FLOP_TRIANGLE * tot["triangle_tests"] +
                      FLOP_LIGHT_SHADE * tot["shaded_hits"] * flat.num_lights)
        # Per-GPU rate of the dominant kernel: each rank traced 1/world of the FLOP.
        achieved = trace_flop / world / (trace_ms * 1e-3) / 1e12 if trace_ms > 0 else 0.0
        # What the trace kernel executed: filter evaluations (4 FMA per ray and entry, 3 FMA +
        # 1 subtraction per origin and entry), the exact Intersect of the candidates (36 FLOP
        # per ray; a record has rays_per_record rays on average) and the plane / triangle tests.
        tr_filter_rays = tot["filter_ray_tests"] - tot["primary_filter_ray_tests"]
        tr_candidates = tot["exact_candidates"] - tot["primary_exact_candidates"]
        rays_per_record = (tot["rays_shadow"] + tot["rays_secondary"]) / max(tot["shaded_hits"], 1.0)
        executed_flop = (8.0 * tr_filter_rays + 7.0 * tot["filter_origin_terms"] +
                         (FLOP_SPHERE + 11.0) * rays_per_record * tr_candidates +
                         FLOP_PLANE * tr_p + FLOP_TRIANGLE * tr_t)
        executed = executed_flop / world / (trace_ms * 1e-3) / 1e12 if trace_ms > 0 else 0.0
        line = {
            "metric": "Mrays/sec (primary+secondary) at 3840x2160",
            "value": round(rays / (ms_per_step * 1e-3) / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "synthetic 1k-sphere scene (1000 spheres + 6 planes, 2 lights), "
                            "3840x2160, 4 mirror bounces",
                "width": WIDTH, "height": HEIGHT, "max_depth": MAX_DEPTH,
                "spheres": ns, "planes": npl, "lights": flat.num_lights,
                "partition": ("whole frame on one GPU" if world == 1 else
                              f"every {world}th band of 16 rows per rank, RCCL gather to rank 0, "
                              "two frames in flight (a frame's gather overlaps the next frame's rendering)"),
                "rays_per_frame": int(rays),
                "exact_candidates_rank0": int(st["exact_candidates"]),
            },
            "roofline": {
                "bound": "valu-fp32", "kernel": "k_trace_levels (every level of the frame in one launch)",
                "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                "traffic": pmc_traffic()[0] if world == 1 else None, "traffic_source": pmc_traffic()[1],
                "algorithmic_flop_per_launch": trace_flop / world,
                "kernel_ms": round(trace_ms, 4),
                "kernel_ms_source": f"HIP events on the kernels' stream, mean of {EVENT_FRAMES} frames after the timed "
                                    "loop, same kernels as the timed loop",
                "executed": {"flop_per_launch": executed_flop / world, "tflops": round(executed, 3),
                             "frac": round(executed / PEAK_FP32_TFLOPS, 4),
                             "filter_ray_tests": int(tr_filter_rays),
                             "algorithmic_sphere_tests": int(tr_s),
                             "note": "bounding-ball culling skips most of the algorithmic sphere tests"},
                "frame": {"flop": frame_flop, "pipeline_ms": round(kernel_ms, 4),
                          "primary_ms": round(primary_ms, 4), "shade_ms": round(shade_ms, 4),
                          "tflops": round(frame_flop / world / (kernel_ms * 1e-3) / 1e12, 3)
                          if kernel_ms > 0 else 0.0},
            },
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(flat, cam)
        print(json.dumps(line), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
