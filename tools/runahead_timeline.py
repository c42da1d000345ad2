#!/usr/bin/env python3
"""Run-ahead passes in rocprofv3's kernel trace: do consecutive passes overlap, and what is the period?

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --n 10000000 --steps 64 --warmup 16 ...
    tools/runahead_timeline.py DIR

Reads DIR/**/*kernel_trace.csv, takes the launches of zf_runahead_kernel in start order and prints one JSON object:
launches, how many start before their predecessor has ended and by how much (the part of the predecessor - its
finalisation - that no longer lies between two passes), the duration of a launch (which includes its wait for the
predecessor's workgroups), and the period between the ENDS of consecutive passes inside a block against the period of
the per-pass kernels of the same trace (warm-up / tail passes), if any."""
import csv
import glob
import json
import os
import statistics
import sys

d = sys.argv[1]
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
ra = [(s, e, q) for s, e, n, q in rows if "zf_runahead_kernel" in n]
out = {"runahead_launches": len(ra), "queues": sorted({q for _, _, q in ra})}
ov, per, dur, first = [], [], [], []
for (s0, e0, q0), (s1, e1, q1) in zip(ra, ra[1:]):
    if s1 - e0 > 200e3:      # another block (a host round trip lies between)
        first.append((e1 - s1) / 1e3)
        continue
    ov.append((e0 - s1) / 1e3)
    per.append((e1 - e0) / 1e3)
    dur.append((e1 - s1) / 1e3)
if ov:
    out.update({
        "pairs_inside_blocks": len(ov),
        "pairs_that_overlap": sum(1 for v in ov if v > 0),
        "overlap_us_median": statistics.median(ov),
        "period_between_ends_us_median": statistics.median(per),
        "period_between_ends_us_mean": statistics.fmean(per),
        "launch_duration_us_median (includes the wait for the predecessor)": statistics.median(dur),
        "first_pass_of_a_block_duration_us_median (nothing to wait for)": statistics.median(first) if first else None,
    })
print(json.dumps(out, indent=1))
