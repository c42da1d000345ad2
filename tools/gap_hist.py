#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 kernel trace (end -> next start), per following kernel name:
    tools/gap_hist.py DIR   (DIR/**/*kernel_trace.csv)"""
import collections
import csv
import glob
import json
import os
import sys

rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
acc = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    acc[n1].append(((s1 - e0) / 1e3, (e1 - s1) / 1e3))
out = {}
for k, v in acc.items():
    g = sorted(x[0] for x in v)
    d = sorted(x[1] for x in v)
    out[k] = {"launches": len(v), "gap_before_us": {"median": g[len(g) // 2], "p10": g[len(g) // 10], "p90": g[len(g) * 9 // 10]},
              "duration_us_median": d[len(d) // 2]}
print(json.dumps(out, indent=1))
