#!/usr/bin/env python3
"""Prints the first N kernel dispatches of a rocprofv3 kernel_trace CSV in time order, and per-kernel totals."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tot = collections.defaultdict(lambda: [0, 0.0])
k = 0
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[name][0] += 1; tot[name][1] += d
    if name.startswith('k_') and k < n:
        print("%-24s %9.1f us  grid %s vgpr %s" % (name[-24:], d, r['Grid_Size_X'], r['VGPR_Count'])); k += 1
print("---- totals")
for name, (c, t) in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%-40s calls %6d total %9.2f ms avg %8.1f us" % (name[:40], c, t / 1e3, t / c))
