#!/usr/bin/env python3
"""Mean duration of the launches of a kernel that did work, from rocprofv3's kernel trace.

A pass of a chained solver is two or three launches of which all but one exit at once (the shape of
the pass lives in the device-resident control block, zf_trial_kernel PART): `--stats` averages the
idle launches in.  This reads <dir>/*kernel_trace.csv and prints, per zf_trial_kernel instance, the
launches, the idle ones (< 50 us) and the mean of the others - the figure bench.py's own HIP events
report as roofline.kernel_avg_ms (its events bracket all launches of a pass: + ~10 us).

    rocprof_busy_mean.py <dir with *kernel_trace.csv> [min_busy_us]"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
floor_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 50e3
out = {}
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        name = row["Kernel_Name"]
        if "zf_trial_kernel" not in name and "zf_runahead_kernel" not in name:
            continue
        dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        e = out.setdefault(name.split("(")[0], {"launches": 0, "idle_launches": 0, "busy_ns": 0, "idle_ns": 0})
        e["launches"] += 1
        if dur < floor_ns:
            e["idle_launches"] += 1
            e["idle_ns"] += dur
        else:
            e["busy_ns"] += dur
for e in out.values():
    busy = e["launches"] - e["idle_launches"]
    e["busy_launches"] = busy
    busy_ns = e.pop("busy_ns")
    e["busy_mean_ms"] = busy_ns / busy / 1e6 if busy else None
    e["idle_mean_us"] = e["idle_ns"] / e["idle_launches"] / 1e3 if e["idle_launches"] else None
    del e["idle_ns"]
print(json.dumps(out, indent=1))
