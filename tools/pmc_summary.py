#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSV rows per kernel and counter.
usage: pmc_summary.py <dir-or-csv> [...]   -> prints one line per (kernel, counter): sum, dispatches, per-dispatch mean"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    files = []
    for a in sys.argv[1:]:
        if os.path.isdir(a):
            files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
        else:
            files.append(a)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                k = k.split("(")[0].replace("void ", "")
                c = row.get("Counter_Name", "?")
                v = float(row.get("Counter_Value", 0) or 0)
                acc[(k, c)][0] += v
                acc[(k, c)][1] += 1
    for (k, c), (s, n) in sorted(acc.items()):
        print(f"{k:28s} {c:28s} sum={s:.6g} dispatches={n} mean={s / max(n, 1):.6g}")


if __name__ == "__main__":
    main()
