#!/usr/bin/env python3
"""Kernels and memory copies of the LAST rt_render call in a rocprofv3 --kernel-trace --memory-copy-trace
database (rocpd sqlite), in start order: start, duration, queue / stream, name.

usage: handoff_timeline.py results.db [gap_us]   (a frame = what follows the last idle gap of gap_us)
"""
import sqlite3
import sys


def rows_of(cur, view, kind):
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    name = "name" if "name" in cols else None
    where = [c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols]
    sel = ", ".join([name or "''", "start", "end"] + where[:2])
    out = []
    for r in cur.execute(f"select {sel} from {view}"):
        out.append((r[1], r[2], kind, str(r[0])[:70], "/".join(str(x) for x in r[3:])))
    return out


def main(path, gap_us=300.0):
    cur = sqlite3.connect(path).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    if "kernels" in tables:
        ev += rows_of(cur, "kernels", "K")
    for t in ("memory_copies", "memory_copy"):
        if t in tables:
            ev += rows_of(cur, t, "C")
            break
    ev.sort()
    # the last frame: after the last gap
    first = 0
    busy_until = ev[0][1]
    for i, e in enumerate(ev):
        if e[0] - busy_until > gap_us * 1e3:
            first = i
        busy_until = max(busy_until, e[1])
    t0 = ev[first][0]
    for s, e, kind, name, where in ev[first:]:
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us {kind} [{where}] {name}")
    print(f"span {(max(e[1] for e in ev[first:]) - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 300.0)
