#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace database (rocpd sqlite) as per-kernel statistics.

usage: rocprof_summary.py results.db [> profiles/name.txt]
Prints, per kernel name: calls, total / average / min / max duration in microseconds and
the share of the summed kernel time — the same table `--stats` writes in CSV mode.
"""
import sqlite3
import sys


def main(path: str) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), "
        f"max(end - start) from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, mn, mx in rows:
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:70s} {calls:6d} {tot / 1e3:12.1f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} "
              f"{100.0 * tot / total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
