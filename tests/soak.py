#!/usr/bin/env python3
"""Random small scenes through the library and the CPU restatement, frame for frame: python tests/soak.py [cases] [seed]
(a checker like the tests beside it, run by hand: pytest does not collect it).
One context for all cases (scenes, sizes and depths change under it; every case renders twice, so that the second frame
runs on the first one's hints).  Prints one line per mismatch and a summary; exit status 1 on any mismatch."""
import os
import sys
import random

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Camera, Light, RenderContext, synthetic_scene  # noqa: E402
from oracle import oracle as O  # noqa: E402  (the checker)
from tests.parity import EXACT_FLOOR, WITHIN1_FLOOR, compare  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.build()
bad = 0
with RenderContext([0]) as c:
    for i in range(cases):
        ns = rng.choice([0, 1, 7, 40, 300, 1024, 1025, 1500, 3000, 9000])
        nl = rng.choice([0, 1, 2, 3, 4, 5, 8, 12, 16, 20])
        depth = rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 10])
        w, h = rng.choice([(64, 36), (96, 54), (160, 90), (200, 113), (320, 180), (33, 17), (480, 270), (640, 360), (960, 540)])
        if ns > 1500 and w * h > 320 * 180:
            w, h = 320, 180
        if ns >= 3000 and w * h > 160 * 90:
            w, h = 160, 90   # (the restatement walks every sphere for every ray)
        if nl >= 12 and depth > 5:
            depth = 5       # (... and retraces the reflection per light)
        sc = synthetic_scene(ns, radius=(0.02, 0.12) if ns < 2000 else (0.01, 0.05), seed=rng.randrange(1 << 30))
        sc.lights.clear()
        for k in range(nl):
            sc.Add(Light(rng.uniform(-3, 3), rng.uniform(-3.5, 0.5), rng.uniform(0.2, 3.3)))
        fs = sc.flatten()
        cam = Camera(position=(rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.3), rng.uniform(-0.3, 0.3)))
        c.set_scene(fs)
        a = c.render(cam, w, h, depth)
        b = c.render(cam, w, h, depth)
        ref, _ = O.render(fs, cam, w, h, max_depth=depth, mode=O.MODE_MEMOISED)
        ok = np.array_equal(a, b)   # (the two frames must be the same frame whatever the restatement says)
        for frame in (a, b):
            r = compare(frame, ref)
            ok = ok and r["exact"] >= EXACT_FLOOR and r["within1"] >= WITHIN1_FLOOR
        if not np.array_equal(a, ref):
            print(f"case {i}: spheres {ns} lights {nl} depth {depth} {w}x{h}: {int((a != ref).sum())} pixels differ from the restatement "
                  f"(largest channel difference {compare(a, ref)['max_diff']}), frames equal: {np.array_equal(a, b)}", flush=True)
        if not ok:
            bad += 1
            print(f"MISMATCH case {i}", flush=True)
        elif i % 10 == 0:
            print(f"case {i}: spheres {ns} lights {nl} depth {depth} {w}x{h} ok", flush=True)
print(f"{cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
