#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (same queue): where the time between the chase
and bulk-apply launches goes.  usage: trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
gaps = defaultdict(list)
for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
    gaps[(k0, k1)].append(s1 - e0)
print("%-28s -> %-28s %8s %10s %10s %10s" % ("after", "before", "count", "avg_ns", "median_ns", "total_ms"))
for (k0, k1), g in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    g.sort()
    print("%-28s -> %-28s %8d %10.0f %10d %10.2f" % (k0, k1, len(g), sum(g) / len(g), g[len(g) // 2], sum(g) / 1e6))
