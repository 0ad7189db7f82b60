#!/usr/bin/env python3
"""bench.py — Mrays/s of the render loop on the 1k-sphere scene at 3840x2160, 4 bounces.

One process per GPU (`python bench.py` for one GPU; the driver launches N ranks through
torch.distributed.run).  A step is one frame: every rank renders its rows of the SAME
frame (every N-th band of 16 rows: rt_render_bands_device; scene and scratch resident in HBM),
then the blocks are gathered on rank 0 with one RCCL collective (torch.distributed backend
"nccl") and put back in row order.  Total work is fixed as N grows: strong scaling.  The timed
frames run without the library's counters; the ray counts and per-kernel HIP-event times come
from a few more frames rendered afterwards.  After the timed loop every rank asks the library
whether the device gave up on any frame (rt_frame_status) and the run fails if one did.

Printed by rank 0: one JSON line with
  value        whole-job Mrays/s = (primary + shadow + used reflection rays per frame,
               de-duplicated as include/rt_hip.h defines them) / seconds per frame; the timed
               step ends with the frame in rank 0's DEVICE memory
  roofline     the levels stage, i.e. everything between the primary-ray kernel and the final
               shading passes: one k_reflect launch per level (the reflection rays' closest-hit
               walks over the LDS-staged scene; k_reflect_grid, one walk per lane through the spheres'
               uniform grid, for scenes of more than 1024 spheres) beside one k_plain launch per level
               on a second stream (light rays and shading of the
               records that reflect nothing; k_shade_lights for short queues of many-light scenes),
               or on small frames one k_trace_levels launch that does both.  The FLOP the stage EXECUTED per frame
               (filter evaluations of rays against spheres and bounding balls - a cell a grid walk
               visits counts as one -, per-origin terms,
               exact Intersect of the candidates and of the spheres the lights' direction tables
               list, plane and triangle tests; from the library's work counters) / its HIP-event
               time, against the 157.3 TFLOP/s vector-FP32 peak.  `algorithmic_tflops` prices the same launch at
               the Intersect calls the reference's brute-force algorithm makes (25 FLOP per
               ray-sphere test ...); the kernel culls most of them, so that figure is a
               speed-up over brute force (`algorithmic_speedup`), not a hardware rate.
               `valu_issue_util`: VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU,
               committed under profiles/) x 2 cycles / (SIMD-cycles of the launch).
  handoff      what the host pays beyond `value`: N = 1, rt_render into pageable host memory
               (Surface.pixels) and into page-locked memory; N > 1, kernel / gather / device-to-
               host times of a frame
  cpu_baseline the CPU restatement (oracle/) timed on the host cores, on a stated sample

`--config c5` runs the 64k-sphere / 16-light / 8-bounce stress scene at 1920x1080 instead.
Rehearsal on a box with fewer GPUs than ranks: RT_BENCH_BACKEND=gloo RT_BENCH_ONE_GPU=1.
"""
from __future__ import annotations

import argparse
import faulthandler
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (imported before librt_hip.so so both share one HIP runtime)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytracer_amd import Camera, RenderContext, RtError, default_scene, scene_1k, scene_64k  # noqa: E402
from raytracer_amd.distributed import FramePipeline, render_block  # noqa: E402

PEAK_FP32_TFLOPS = 157.3          # 256 CUs x 4 SIMDs x 64 lanes x 2 FLOP / 2 cycles x 2.4 GHz
SIMDS, CLOCK_HZ, VALU_ISSUE_CYCLES = 1024, 2.4e9, 2.0
FLOP_SPHERE, FLOP_PLANE, FLOP_TRIANGLE, FLOP_LIGHT_SHADE = 25.0, 13.0, 51.0, 80.0

CONFIGS = {
    # The reference's own scene, primary + shadow rays only (max_depth 0: no mirror recursion).
    "c2": dict(width=1920, height=1080, max_depth=0, scene=default_scene, cpu_row_step=1, cpu_mode="literal",
               workload="reference default scene (2 spheres, 1 triangle, 6 planes, 2 lights), 1920x1080, primary + shadow rays only",
               pmc=None),
    # The configuration the headline metric is quoted on.
    "c3": dict(width=3840, height=2160, max_depth=4, scene=scene_1k, cpu_row_step=1, cpu_mode="literal",
               workload="synthetic 1k-sphere scene (1000 spheres + 6 planes, 2 lights), 3840x2160, 4 mirror bounces",
               pmc="r03_c3_trace_pmc.json"),
    # The same scene at 7680x4320: the frame the row partition over 8 GPUs is specified on (one GPU renders all of it).
    "c4": dict(width=7680, height=4320, max_depth=4, scene=scene_1k, cpu_row_step=8, cpu_mode="literal",
               workload="synthetic 1k-sphere scene (1000 spheres + 6 planes, 2 lights), 7680x4320, 4 mirror bounces",
               pmc=None),
    # Divergence / occupancy stress.
    "c5": dict(width=1920, height=1080, max_depth=8, scene=scene_64k, cpu_row_step=54, cpu_mode="memoised",
               workload="synthetic 64k-sphere scene (65536 spheres + 6 planes, 16 lights), 1920x1080, 8 mirror bounces",
               pmc="r03_c5_trace_pmc.json"),
}
STEADY_FRAMES = 300  # frames of the untimed loop behind a short timed loop (`ms_per_frame_of_300_more_frames`)
WARM_SECONDS = 0.3   # untimed frames before the timed loop run at least this long: the clocks have ramped, whatever --steps


def cpu_baseline(cfg, flat, cam) -> dict:
    """CPU path next to the GPU number, rows spread over the host threads like Parallel.For.
    Literal mode re-traces the reflection per light as objects.cs:80-88 does; with 16 lights and
    8 bounces that is 16^8 re-traces, so that scene is timed in memoised mode (same pixels)."""
    from oracle import oracle as O  # checker / baseline only; never on the product path
    O.build()
    w, h, depth, step = cfg["width"], cfg["height"], cfg["max_depth"], cfg["cpu_row_step"]
    mode = O.MODE_LITERAL if cfg["cpu_mode"] == "literal" else O.MODE_MEMOISED
    threads = O.num_threads()
    t0 = time.perf_counter()
    passes = 0
    while passes == 0 or (time.perf_counter() - t0 < 3.0 and passes < 1000):   # (a small frame: again until it is a measurement)
        _, cnt = O.render(flat, cam, w, h, max_depth=depth, mode=mode, row_step=step)
        passes += 1
    dt = (time.perf_counter() - t0) / passes
    if mode != O.MODE_MEMOISED:   # ray count of the same rows, de-duplicated like the GPU's (not timed)
        _, cnt = O.render(flat, cam, w, h, max_depth=depth, mode=O.MODE_MEMOISED, row_step=step)
    rays = cnt["rays_primary"] + cnt["rays_shadow"] + cnt["rays_secondary"]
    how = "literal per-light recursion" if mode == O.MODE_LITERAL else "memoised recursion (one trace per level)"
    return {
        "value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"{len(range(0, h, step))} of the {h} rows (every {step}) of the {w}x{h} frame, {how}, "
                  f"{passes} pass{'es' if passes > 1 else ''} of {dt:.2f} s",
    }


def host_handoff(config: str) -> dict:
    """rt_render as a host WITHOUT PyTorch sees it - the C# application loads librt_hip.so against the
    system's HIP runtime, whose device-to-host copies run on the DMA engines; the runtime inside the PyTorch
    wheel, which this process shares with torch, makes them with a shader kernel that takes the device from
    whatever renders beside it.  tools/handoff_host.py in a process of its own, started BEFORE this one touches
    the GPU."""
    import subprocess
    env = dict(os.environ, RT_HIP_SYSTEM_RUNTIME="1", PROBE_REPS="20")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "handoff_host.py"), config], env=env,
                           capture_output=True, text=True, timeout=180)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        row = d["rows"][0]
        if not (row["identical"] and row["pageable_identical"]):
            return {"error": "frames differ between the hand-off paths"}
        return {"rt_render_library_framebuffer_ms": row["library_framebuffer_ms"], "rt_render_pageable_ms": row["pageable_ms"],
                "kernels_alone_ms": d["kernels_ms"], "runtime": "system HIP runtime (/opt/rocm), no PyTorch in the process"}
    except Exception as e:   # noqa: BLE001  (a missing tool or a failed child must not cost the bench line)
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def pmc_summary(name) -> dict:
    """Per-frame counters of the levels stage on this workload from the committed rocprofv3
    PMC passes (they cannot be collected from inside this process); {} when missing."""
    if not name:
        return {}
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        d["source"] = f"profiles/{name} (rocprofv3 --pmc, separate passes, one GPU)"
        return d
    except (OSError, ValueError):
        return {}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-handoff", action="store_true", help="skip the host hand-off measurements (N = 1)")
    ap.add_argument("--partition", choices=["interleaved", "contiguous"], default="interleaved",
                    help="N > 1: every N-th band of 16 rows per rank, or one block of height / N rows")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    W, H, DEPTH = cfg["width"], cfg["height"], cfg["max_depth"]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("RT_BENCH_ONE_GPU") else local_rank
    # (before anything here touches the GPU: the child has the device to itself)
    host_side = host_handoff(args.config) if world == 1 and not args.no_handoff else None
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        # A rank that never arrives must end the job with a traceback, not hold the node.
        faulthandler.dump_traceback_later(int(os.environ.get("RT_BENCH_WATCHDOG_S", "300")), exit=True)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        nranks = dist.get_world_size()
    else:
        dist = None
        nranks = 1

    cam = Camera()
    flat = cfg["scene"]().flatten()
    # One GPU: one frame at a time (what the C# host's blocking Render() sees, and the condition
    # the kernel durations below are measured under).  Several ranks: frames in flight, each in a
    # render context of its own, so that a frame's gather (and rank 0's row permutation) and the
    # idle stretches of its chain of reflection levels are filled by the next frames' kernels (four:
    # measured on one MI355X with a rank's share of the frame, profiles/r02_pipeline_sweep.txt).
    in_flight = int(os.environ.get("RT_BENCH_FRAMES_IN_FLIGHT", "0")) or (1 if world == 1 else 4)
    contexts = [RenderContext([dev_index]) for _ in range(in_flight)]
    for c in contexts:
        c.set_scene(flat)
        c.set_device_share(in_flight)
    ctx = contexts[0]
    interleave = args.partition == "interleaved"
    pipeline = FramePipeline(W, H, rank, world, device, depth=in_flight, contexts=contexts, interleave=interleave)

    def step() -> None:
        pipeline.render_frame(None, cam, DEPTH)

    def fence() -> None:
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # ... and on until the device has worked for WARM_SECONDS (every rank the same number of frames: collectives)
    t_warm = time.perf_counter()
    for _ in range(3):
        step()
    fence()
    per_frame = max((time.perf_counter() - t_warm) / 3, 1e-5)
    more = torch.tensor([max(0.0, WARM_SECONDS / per_frame - 3)], dtype=torch.float64)
    if dist is not None:
        more = more.cuda() if backend == "nccl" else more
        dist.all_reduce(more, op=dist.ReduceOp.MAX)
    extra_warm = int(more.item())
    for _ in range(extra_warm):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # A short timed loop carries what starting from an idle device costs (the first launch after the fence finds the
    # queues asleep: ~0.3 ms, i.e. 4 % of twenty 4K frames); the same loop over STEADY_FRAMES frames right behind it,
    # for the record next to `ms_per_step`, never instead of it.
    steady_ms = None
    if args.steps < STEADY_FRAMES:
        t1 = time.perf_counter()
        for _ in range(STEADY_FRAMES):
            step()
        fence()
        steady_ms = (time.perf_counter() - t1) * 1e3 / STEADY_FRAMES

    # Did the device finish every frame it was given?  (A stalled level schedule leaves a frame
    # incomplete; its time must not be reported.)  Every rank checks, every rank learns the result.
    stalled = 0.0
    for c in contexts:
        try:
            c.frame_status()
        except RtError as e:
            print(f"[bench rank {rank}] {e}", file=sys.stderr, flush=True)
            stalled = 1.0
    for c in contexts:   # what follows measures one frame at a time on the first context
        c.set_device_share(1)

    # Kernel durations of the SAME kernels the timed loop ran: a few more frames bracketed by HIP
    # events on the stream (profiling mode 2), averaged.
    ctx.set_profiling(2)
    ev = {"trace_ms": 0.0, "kernel_ms": 0.0, "primary_ms": 0.0, "shade_ms": 0.0}
    EVENT_FRAMES = 5
    for _ in range(EVENT_FRAMES):
        render_block(ctx, cam, W, H, DEPTH, pipeline.slots[0])
        timed = ctx.stats()  # synchronises
        for k in ev:
            ev[k] += timed[k] / EVENT_FRAMES
    # One more frame with the device-side work counters on (profiling mode 1: the counting
    # instances of the kernels, a few per cent slower; only their counts are used).
    ctx.set_profiling(1)
    render_block(ctx, cam, W, H, DEPTH, pipeline.slots[0])
    st = ctx.stats()
    ctx.set_profiling(0)

    # N > 1, what a BLOCKING Render() sees (MyApplication.cs:179-193 presents every frame before it starts the
    # next): one frame at a time - this rank's rows, the gather, the row permutation on rank 0, all ranks in
    # step - next to the frames-in-flight figure of the timed loop; and the same with the other partition.
    blocking = {}
    if world > 1:
        def one_at_a_time(pipe, frames=20) -> float:
            for _ in range(3):
                pipe.render_frame(ctx, cam, DEPTH)
                fence()
            t = time.perf_counter()
            for _ in range(frames):
                pipe.render_frame(ctx, cam, DEPTH)
                fence()
            return (time.perf_counter() - t) / frames * 1e3
        single = FramePipeline(W, H, rank, world, device, depth=1, contexts=[ctx], interleave=interleave)
        other = FramePipeline(W, H, rank, world, device, depth=1, contexts=[ctx], interleave=not interleave)
        blocking = {"one_frame_at_a_time_ms": one_at_a_time(single),
                    ("contiguous" if interleave else "interleaved") + "_partition_one_frame_at_a_time_ms": one_at_a_time(other)}
        del single, other

    # Beyond the timed step: N > 1, the gather and the copy of the gathered frame to the host, each
    # timed on its own; N = 1 below (rank 0 only).
    handoff = {}
    if world > 1:
        g = pipeline.slots[0]
        render_block(ctx, cam, W, H, DEPTH, g)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(g.stream):
            e0.record()
        g.gather()
        host = torch.empty((H, W), dtype=torch.int32, pin_memory=True) if rank == 0 else None
        with torch.cuda.stream(g.stream):
            e1.record()
            if rank == 0:
                host.copy_(g.frame, non_blocking=True)
            e2.record()
        g.stream.synchronize()
        handoff = {"kernel_ms": round(ev["kernel_ms"], 4), "gather_ms": round(e0.elapsed_time(e1), 4),
                   "d2h_ms": round(e1.elapsed_time(e2), 4) if rank == 0 else None,
                   "note": "rank 0; one frame after the timed loop: this rank's kernels (HIP events), the gather "
                           f"({backend}) with rank 0's row permutation, the copy of the whole frame into page-locked "
                           "host memory; in the timed loop the gather overlaps the next frame's kernels"}

    keys = ("rays_primary", "rays_shadow", "rays_secondary", "shaded_hits", "sphere_tests",
            "plane_tests", "triangle_tests", "filter_ray_tests", "filter_origin_terms", "exact_candidates",
            "primary_filter_ray_tests", "primary_exact_candidates", "shadow_candidates")
    totals = torch.tensor([float(st[k]) for k in keys], dtype=torch.float64, device="cuda")
    times = torch.tensor([elapsed, ev["trace_ms"], ev["kernel_ms"], ev["primary_ms"], ev["shade_ms"], stalled, steady_ms or 0.0] +
                         list(blocking.values()), dtype=torch.float64, device="cuda")
    if dist is not None:
        if backend != "nccl":   # (host tensors: see FrameGatherer._gather_through_host)
            totals, times = totals.cpu(), times.cpu()
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    tot = dict(zip(keys, totals.tolist()))
    elapsed, trace_ms, kernel_ms, primary_ms, shade_ms, stalled, steady_ms = times.tolist()[:7]
    blocking = dict(zip(blocking, (round(v, 4) for v in times.tolist()[7:])))
    if stalled:
        for c in contexts:
            c.close()
        raise SystemExit("bench: the device gave up on a frame of the timed loop (see stderr); no result")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays = tot["rays_primary"] + tot["rays_shadow"] + tot["rays_secondary"]
        ns, npl, nt = (int((flat.prim_types == t).sum()) for t in (0, 1, 2))
        # Tests made by the traversal kernel = all tests minus the primary rays' share.
        tr_s = tot["sphere_tests"] - tot["rays_primary"] * ns
        tr_p = tot["plane_tests"] - tot["rays_primary"] * npl
        tr_t = tot["triangle_tests"] - tot["rays_primary"] * nt
        algorithmic_flop = FLOP_SPHERE * tr_s + FLOP_PLANE * tr_p + FLOP_TRIANGLE * tr_t
        # What the traversal kernel executed.  Reflection rays: filter evaluations against spheres
        # and bounding balls (4 FMA per ray and entry, 3 FMA + 1 subtraction per origin and entry)
        # and the exact Intersect of the candidates (25 FLOP to the discriminant + 11 to the hit).
        # Light rays: the exact Intersect of every sphere their light's direction table lists
        # (25 FLOP to the discriminant; few go on).  Plane / triangle tests as the reference makes them.
        # The records that reflect nothing are shaded inside the stage (80 FLOP per record and light).
        tr_filter_rays = tot["filter_ray_tests"] - tot["primary_filter_ray_tests"]
        tr_candidates = tot["exact_candidates"] - tot["primary_exact_candidates"]
        executed_flop = (8.0 * tr_filter_rays + 7.0 * tot["filter_origin_terms"] +
                         (FLOP_SPHERE + 11.0) * tr_candidates + FLOP_SPHERE * tot["shadow_candidates"] +
                         FLOP_PLANE * tr_p + FLOP_TRIANGLE * tr_t +
                         FLOP_LIGHT_SHADE * flat.num_lights * (tot["shaded_hits"] - tot["rays_secondary"]))
        # Scenes of at most four lights, one frame at a time: the reflecting records' light terms are computed
        # inside the stage as well (k_lit, beside the plain records' launches), one reflecting record per secondary ray.
        if flat.num_lights <= 4 and DEPTH > 0 and in_flight == 1:
            executed_flop += FLOP_LIGHT_SHADE * flat.num_lights * tot["rays_secondary"]
        trace_s = trace_ms * 1e-3
        # Per-GPU rate of the dominant kernel: each rank traced 1/world of the FLOP.
        executed = executed_flop / world / trace_s / 1e12 if trace_s > 0 else 0.0
        algorithmic = algorithmic_flop / world / trace_s / 1e12 if trace_s > 0 else 0.0
        pmc = pmc_summary(cfg["pmc"]) if world == 1 else {}
        valu_insts = pmc.get("sq_insts_valu_per_launch")
        line = {
            "metric": "Mrays/sec (primary+secondary) at 3840x2160" if args.config == "c3"
                      else f"Mrays/sec (primary+secondary) at {W}x{H}",
            "value": round(rays / (ms_per_step * 1e-3) / 1e6, 2),
            "unit": "Mrays/s",
            # `value` counts the light (shadow) rays too, as the survey defines the metric; this is the same
            # frame rate priced at the primary and reflection (secondary) rays alone.
            "Mrays_primary_plus_secondary": round((tot["rays_primary"] + tot["rays_secondary"]) / (ms_per_step * 1e-3) / 1e6, 2),
            "n_gpus": world, "nranks": nranks, "steps": args.steps, "warmup": args.warmup,
            "untimed_frames_before_the_loop": args.warmup + 3 + extra_warm,
            "ms_per_step": round(ms_per_step, 4),
            # (short loops only: the same loop over 300 more frames right behind the timed one; the timed loop starts from
            # an idle device, which costs ~0.3 ms whatever the number of steps)
            "ms_per_frame_of_300_more_frames": round(steady_ms, 4) if steady_ms else None,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": cfg["workload"],
                "width": W, "height": H, "max_depth": DEPTH,
                "spheres": ns, "planes": npl, "lights": flat.num_lights,
                "partition": ("whole frame on one GPU" if world == 1 else
                              (f"every {world}th band of 16 rows per rank" if interleave else f"one block of {H}/{world} rows per rank") +
                              f", {backend} gather to rank 0, "
                              f"{in_flight} frames in flight per rank, each in a render context of its own (a frame's "
                              "gather and the idle stretches of its level chain overlap the next frames' kernels)"),
                "frames_in_flight": in_flight,
                "rays_counted": "primary + shadow + secondary (one shadow ray per shaded hit and light, one "
                                "secondary ray per shaded specular hit below the depth bound, de-duplicated per level)",
                "rays_per_frame": {"primary": int(tot["rays_primary"]), "shadow": int(tot["rays_shadow"]),
                                   "secondary": int(tot["rays_secondary"]), "total": int(rays)},
                "counts_source": "one frame after the timed loop rendered with the library's work counters on "
                                 "(rt_set_profiling 1); kernel times: five frames with HIP events (mode 2)",
                "timed_step_ends": "frame in device memory of rank 0 (see handoff for the host copy)",
            },
            "roofline": {
                "bound": "valu-fp32", "kernel": "levels stage of a frame: k_reflect (many-chunk scenes: k_reflect_grid) per level (reflection "
                                                 "rays) on a second stream beside k_plain (or k_shade_lights) per level (light rays + shading "
                                                 "of plain records); 'launch' = one frame's stage",
                "achieved": round(executed, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(executed / PEAK_FP32_TFLOPS, 4),
                "executed_flop_per_launch": executed_flop / world,
                "kernel_ms": round(trace_ms, 4),
                "kernel_ms_source": f"HIP events on the kernels' stream, mean of {EVENT_FRAMES} frames after the timed "
                                    "loop, same kernels as the timed loop",
                "valu_issue_util": (round(valu_insts * VALU_ISSUE_CYCLES / (trace_s * SIMDS * CLOCK_HZ), 4)
                                    if valu_insts and trace_s > 0 else None),
                "valu_insts_per_launch": valu_insts,
                "traffic": pmc.get("traffic_bytes_per_launch"), "pmc_source": pmc.get("source"),
                "frac_basis": "FLOP the stage EXECUTED.  Priced at the Intersect calls the reference's brute-force loops make "
                              "(25 FLOP per ray-sphere test ...), the same launch comes to `algorithmic_tflops`, several times "
                              "the peak: the kernels never make 97 % of those tests (culling), so that figure is a speed-up "
                              "over brute force (`algorithmic_speedup`), not a rate of the hardware, and is not used for `frac`",
                "algorithmic_tflops": round(algorithmic, 3),
                "algorithmic_flop_per_launch": algorithmic_flop / world,
                "algorithmic_speedup": round(algorithmic_flop / executed_flop, 3) if executed_flop > 0 else None,
                "filter_ray_tests": int(tr_filter_rays), "shadow_candidates": int(tot["shadow_candidates"]),
                "reflection_candidates": int(tr_candidates), "algorithmic_sphere_tests": int(tr_s),
                "frame": {"pipeline_ms": round(kernel_ms, 4), "primary_ms": round(primary_ms, 4),
                          "shade_ms": round(shade_ms, 4)},
            },
        }
        if world == 1:
            # Render() ends with the pixels in host memory (MyApplication.cs:420): rt_render into a
            # pageable array, and into a page-locked one where the system allows registering it.
            out = np.empty((H, W), dtype=np.int32)
            reps = 10

            def host_ms() -> float:
                for _ in range(5):   # (staging buffers, events, helper threads and the library's own streams are set up by the first calls)
                    ctx.render(cam, W, H, DEPTH, out=out)
                t = time.perf_counter()
                for _ in range(reps):
                    tt = time.perf_counter()
                    ctx.render(cam, W, H, DEPTH, out=out)
                    if os.environ.get("RT_BENCH_DEBUG"):
                        print(f"[bench] rt_render {(time.perf_counter() - tt) * 1e3:.3f} ms", file=sys.stderr, flush=True)
                return (time.perf_counter() - t) / reps * 1e3

            pageable = host_ms()
            try:
                ctx.pin_pixels(out)
                pinned = round(host_ms(), 4)
                ctx.unpin_pixels()
            except RtError as e:
                pinned = f"refused ({e.message.split(';')[0]})"
            # ... and into a page-locked framebuffer of the library's own (rt_alloc_pixels): the direct DMA,
            # no staging and no move by the caller's thread.
            out = ctx.alloc_pixels(H, W)
            owned = round(host_ms(), 4)
            ctx.unpin_pixels()
            # The same frames with two in flight, each in a render context of its own (a host that
            # presents from device memory can enqueue frame i + 1 before frame i is done): the idle
            # stretches of one frame's chain of levels are filled by the other's kernels.
            second = RenderContext([dev_index])
            second.set_scene(flat)
            pair = [ctx, second]
            for c in pair:
                c.set_device_share(2)
            two = FramePipeline(W, H, rank, world, device, depth=2, contexts=pair)
            reps2 = max(10, min(200, args.steps // 2))
            for _ in range(4):
                two.render_frame(None, cam, DEPTH)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(reps2):
                two.render_frame(None, cam, DEPTH)
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t2) / reps2 * 1e3
            for c in pair:
                c.frame_status()
                c.set_device_share(1)
            second.close()
            line["pipelined"] = {"frames_in_flight": 2, "ms_per_frame": round(ms2, 4),
                                 "Mrays_per_s": round(rays / (ms2 * 1e-3) / 1e6, 2),
                                 "note": "not `value`: two render contexts on the one GPU, frames enqueued back to back "
                                         "without waiting; `value` is one frame at a time, what a blocking Render() sees"}
            # The floor of any hand-off: ONE copy of the finished frame, device -> page-locked host memory, alone.
            dev_frame = pipeline.slots[0].frame
            pinned_frame = torch.empty((H, W), dtype=torch.int32, pin_memory=True)
            for _ in range(3):
                pinned_frame.copy_(dev_frame, non_blocking=True)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10):
                pinned_frame.copy_(dev_frame, non_blocking=True)
                torch.cuda.synchronize()
            d2h_only = (time.perf_counter() - t) / 10 * 1e3
            del pinned_frame
            handoff = {"d2h_only_ms": round(d2h_only, 4), "d2h_GB_per_s": round(W * H * 4 / (d2h_only * 1e-3) / 1e9, 2),
                       "host_without_pytorch": host_side,
                       "rt_render_pageable_ms": round(pageable, 4), "rt_render_pinned_ms": pinned,
                       "rt_render_library_framebuffer_ms": owned,
                       "note": "whole rt_render calls (kernels + copy into the caller's int[w*h]); a frame of 3 M pixels or more "
                               "is rendered as row blocks (interleaved bands) whose copies overlap the next block's kernels.  "
                               "`host_without_pytorch`: the same calls from a process that links the system's HIP runtime, like "
                               "the C# host (copies on the DMA engines); the other rt_render figures: this process, on the "
                               "PyTorch wheel's runtime (copies by a shader kernel)"}
        if blocking:
            line["blocking"] = dict(blocking, Mrays_per_s=round(rays / (blocking["one_frame_at_a_time_ms"] * 1e-3) / 1e6, 2),
                                    note="not `value`: one frame at a time (every rank renders its rows, gather, row permutation, "
                                         "all ranks in step before the next frame starts) - what a blocking Render() sees on N "
                                         "ranks; `value` keeps frames in flight")
        line["handoff"] = handoff
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, flat, cam)
        print(json.dumps(line), flush=True)

    for c in contexts:
        c.close()
    if dist is not None:
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
