#!/usr/bin/env python3
"""rt_render as a host WITHOUT PyTorch sees it (the C# application): numpy + ctypes only, the system's HIP runtime.

handoff_host.py [c3|c4|c2|c5] — prints one JSON line: ms per blocking rt_render call into the library's page-locked
framebuffer and into pageable memory, frames identical or not.  PROBE_SWEEP="parts,bands;..." sweeps the
row-block knobs (RT_HIP_FRAME_PARTS, RT_HIP_PART_BANDS); RT_HIP_SYSTEM_RUNTIME=0 in the environment keeps the PyTorch wheel's runtime (for comparison).
"""
import json
import os
import sys
import time

if os.environ.get("RT_HIP_SYSTEM_RUNTIME", "1") == "0":
    os.environ.pop("RT_HIP_SYSTEM_RUNTIME")
else:
    os.environ["RT_HIP_SYSTEM_RUNTIME"] = "1"

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, RenderContext, scene_1k, scene_64k, default_scene  # noqa: E402

assert "torch" not in sys.modules
REPS = int(os.environ.get("PROBE_REPS", "30"))


def timed(fn, reps=REPS, warm=5):
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    W, H, D, scene = {"c3": (3840, 2160, 4, scene_1k), "c4": (7680, 4320, 4, scene_1k),
                      "c2": (1920, 1080, 0, default_scene), "c5": (1920, 1080, 8, scene_64k)}[which]
    cam = Camera()
    ctx = RenderContext([0])
    ctx.set_scene(scene().flatten())
    ctx.set_profiling(2)               # one piece, kernels then copy: the reference frame and the kernels' own time
    ref = ctx.render(cam, W, H, D).copy()
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        ctx.render(cam, W, H, D, out=ref)
    st = ctx.stats()
    ctx.set_profiling(0)
    out = {"config": which, "frame": f"{W}x{H}", "runtime": "system" if os.environ.get("RT_HIP_SYSTEM_RUNTIME") else "torch wheel",
           "kernels_ms": round(st["kernel_ms"], 4), "copy_after_kernels_ms": round(st["copy_ms"], 4), "rows": []}
    sweep = os.environ.get("PROBE_SWEEP", "")
    settings = [tuple(x.split(",")) for x in sweep.split(";") if x] or [("", "")]
    pageable = np.empty((H, W), dtype=np.int32)
    for setting in settings:
        setting = tuple(setting) + ("",) * (2 - len(setting))
        for k, v in zip(("RT_HIP_FRAME_PARTS", "RT_HIP_PART_BANDS"), setting):
            if v:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        ctx.reload_tuning()
        owned = ctx.alloc_pixels(H, W)
        owned[:] = 0
        row = {"parts": setting[0], "bands": setting[1]}
        row["library_framebuffer_ms"] = round(timed(lambda: ctx.render(cam, W, H, D, out=owned)), 4)
        row["identical"] = bool((owned == ref).all())
        ctx.unpin_pixels()
        pageable[:] = 0
        row["pageable_ms"] = round(timed(lambda: ctx.render(cam, W, H, D, out=pageable)), 4)
        row["pageable_identical"] = bool((pageable == ref).all())
        out["rows"].append(row)
        print(row, file=sys.stderr, flush=True)
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
