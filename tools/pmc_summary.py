#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per kernel name and counter.

usage: pmc_summary.py counter_collection.csv
"""
import csv
import sys
from collections import defaultdict


def main(path):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
    counters = sorted({c for k in acc for c in acc[k]})
    print(f"{'kernel':40s} {'calls':>5s} " + " ".join(f"{c:>20s}" for c in counters))
    for k in sorted(acc, key=lambda k: -acc[k].get(counters[0], 0)):
        print(f"{k[:40]:40s} {len(calls[k]):5d} " + " ".join(f"{acc[k].get(c, 0):20.4g}" for c in counters))


if __name__ == "__main__":
    main(sys.argv[1])
