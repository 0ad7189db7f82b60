#!/usr/bin/env python3
"""Per-launch kernel durations of the last frame in a rocprofv3 --kernel-trace database (rocpd
sqlite), in launch order: with RT_HIP_LEVEL_KERNELS=1 that is one line per queue per level.

usage: level_times.py results.db [launches]
"""
import sqlite3
import sys


def main(path: str, last: int) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in tables else next(t for t in tables if "kernel" in t.lower())
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, start, end from {view} order by start").fetchall()
    rows = rows[-last:]
    t0 = rows[0][1]
    for name, start, end in rows:
        print(f"{(start - t0) / 1e3:10.1f} us  +{(end - start) / 1e3:9.1f} us  {name[:90]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16)
