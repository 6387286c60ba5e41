#!/usr/bin/env python3
"""Concurrency digest of a rocprofv3 kernel trace: per kernel total time, wall span, time with >=2 kernels running."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
tot = collections.defaultdict(float)
for r in rows:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not name.startswith('k_'):
        continue
    tot[name] += (b - a) / 1e6
    ev.append((a, 1, name)); ev.append((b, -1, name))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
depth = 0; last = t0; hist = collections.defaultdict(float)
running = collections.Counter(); ext_alone = 0.0; ext_with_other = 0.0
for t, d, name in ev:
    hist[depth] += (t - last) / 1e6
    if running['ext'] > 0:
        if depth > running['ext']: ext_with_other += (t - last) / 1e6
        else: ext_alone += (t - last) / 1e6
    last = t; depth += d
    if name.startswith('k_extend'): running['ext'] += d
print("span %.1f ms" % ((t1 - t0) / 1e6))
for k in sorted(hist): print("  %d kernels running: %.1f ms" % (k, hist[k]))
print("  traversal running alone %.1f ms, with another kernel %.1f ms" % (ext_alone, ext_with_other))
for n, v in sorted(tot.items(), key=lambda x: -x[1]): print("  %-40s %.1f ms" % (n[:40], v))
