#!/usr/bin/env python3
"""Per-frame figures of the levels stage (the launches between k_primary and the first k_shade of a
frame: k_reflect / k_reflect_chunks / k_reflect_finish and k_plain / k_shade_lights per level, or one
k_trace_levels) from rocprofv3 output.  (k_shade_lights also runs beside k_shade after the stage; its
launches there are counted with the stage's sums but do not stretch the stage's span.)

  stage_profile.py span results.db [frames]
      kernel-trace database (rocpd sqlite): per frame the wall span of the stage (first launch's
      start to last launch's end: the two streams' launches overlap, so the span is shorter than
      the sum of the durations), the summed durations per kernel, and the whole frame's span;
      `frames` = how many frames from the start of the trace to use (default: all).
  stage_profile.py pmc out.json name=counter_collection.csv ...
      counter passes (csv, one file per pass): counters summed over the stage's launches, divided
      by the number of frames; FETCH_SIZE doubled (gfx950 tallies its 128-byte read requests as
      64 bytes) and both sizes converted from KiB to bytes.
"""
import csv
import json
import sqlite3
import sys
from collections import defaultdict

STAGE = ("k_reflect", "k_plain", "k_trace_levels", "k_shade_lights", "k_lit", "k_bin_")


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name.split("(")[0]


def span(path: str, frames: int) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    per_frame, cur_frame = [], None
    for name, start, end in rows:
        k = short(name)
        if k.startswith("k_primary"):
            cur_frame = []
            per_frame.append(cur_frame)
        if cur_frame is not None and k.startswith(("k_primary", "k_shade") + STAGE):
            cur_frame.append((k, start, end))
    if frames > 0:
        per_frame = per_frame[:frames]
    stage_span, frame_span, sums = [], [], defaultdict(list)
    for fr in per_frame:
        first_shade = min([s for k, s, e in fr if k.startswith("k_shade<")], default=None)
        st = [(k, s, e) for k, s, e in fr if k.startswith(STAGE) and (first_shade is None or s < first_shade)]
        if not st:
            continue
        stage_span.append(max(e for _, _, e in st) - min(s for _, s, _ in st))
        frame_span.append(max(e for _, _, e in fr) - min(s for _, s, _ in fr))
        tot = defaultdict(float)
        for k, s, e in fr:
            tot[k] += e - s
        for k, v in tot.items():
            sums[k].append(v)
    n = len(stage_span)
    mean = lambda xs: sum(xs) / max(1, len(xs))   # noqa: E731
    print(f"frames: {n}")
    print(f"levels stage, wall span per frame: mean {mean(stage_span) / 1e3:9.1f} us   min {min(stage_span) / 1e3:9.1f}"
          f"   max {max(stage_span) / 1e3:9.1f}")
    print(f"whole frame (k_primary .. last k_shade): mean {mean(frame_span) / 1e3:9.1f} us")
    print("summed launch durations per frame, by kernel (launches of the two streams overlap):")
    for k in sorted(sums, key=lambda k: -mean(sums[k])):
        print(f"  {k:60s} {mean(sums[k]) / 1e3:10.1f} us")


def pmc(out: str, passes) -> None:
    result = {"kernels": "the launches of a frame between k_primary and the first k_shade: k_reflect / k_reflect_chunks / "
                         "k_reflect_finish, k_plain / k_plain_tail / k_shade_lights / k_lit / k_bin_* (or k_trace_levels)", "per": "frame"}
    for spec in passes:
        label, path = spec.split("=", 1)
        acc, frames = defaultdict(float), set()
        per_kernel = defaultdict(lambda: defaultdict(float))
        with open(path) as f:
            rows = sorted(csv.DictReader(f), key=lambda row: int(row["Dispatch_Id"]))
        in_stage = False   # between a frame's k_primary and its first k_shade
        for row in rows:
            k = short(row["Kernel_Name"])
            if k.startswith("k_primary"):
                frames.add(row["Dispatch_Id"])
                in_stage = True
            elif k.startswith("k_shade<"):
                in_stage = False
            if in_stage and k.startswith(STAGE):
                acc[row["Counter_Name"]] += float(row["Counter_Value"])
                per_kernel[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n = max(1, len(frames))
        result.setdefault("frames_profiled", {})[label] = n
        for c, v in acc.items():
            result[c] = v / n
        for k, d in per_kernel.items():
            for c, v in d.items():
                result.setdefault("by_kernel", {}).setdefault(k, {})[c] = v / n
    if "SQ_INSTS_VALU" in result:
        result["sq_insts_valu_per_launch"] = result["SQ_INSTS_VALU"]
    if "FETCH_SIZE" in result and "WRITE_SIZE" in result:
        result["fetch_bytes_per_launch"] = result["FETCH_SIZE"] * 1024.0 * 2.0
        result["write_bytes_per_launch"] = result["WRITE_SIZE"] * 1024.0
        result["traffic_bytes_per_launch"] = result["fetch_bytes_per_launch"] + result["write_bytes_per_launch"]
    result["note"] = ("rocprofv3 --pmc, one pass per counter group, kernels serialised by the profiler; a 'launch' here "
                      "is the frame's whole levels stage; FETCH_SIZE doubled (gfx950 tallies 128-byte read requests "
                      "as 64 bytes), sizes in KiB converted to bytes")
    with open(out, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "span":
        span(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    else:
        pmc(sys.argv[2], sys.argv[3:])
