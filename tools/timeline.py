#!/usr/bin/env python3
"""Where the time of a chained solve goes, pass by pass, from rocprofv3's kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/bench_configs.py --cfg 2 ...
    tools/timeline.py DIR [first_busy_trial_launch_to_skip]

Reads DIR/**/*kernel_trace.csv, orders the launches by start time and prints one JSON object: per kernel
class (trial kernel busy / idle, finalize, decide, everything else) the launches, mean duration and mean
GAP in front of it (end of the previous kernel -> its start), and the accounting of the span between the
first and the last busy trial launch: kernel time by class, idle gaps, span per busy pass; and the time between
consecutive busy passes of one solve (end of one trial kernel -> start of the next, everything in between included)."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
LIST = "--list" in sys.argv
args = [a for a in sys.argv[2:] if not a.startswith("--")]
skip = int(args[0]) if args else 0
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def cls(name, dur):
    if "zf_trial_kernel" in name or "zf_pass_kernel" in name:
        return "trial_busy" if dur >= 20e3 else "trial_idle"
    for key in ("zf_finalize_kernel", "zf_decide_kernel", "zf_refresh_beta", "zf_set_max_iter", "zf_init", "zf_eval_kernel"):
        if key in name:
            return key
    return "other"


busy = [i for i, (s, e, n) in enumerate(rows) if cls(n, e - s) == "trial_busy"]
if LIST:   # every busy trial launch: the PART of the kernel (last integer template argument) and its duration
    import re

    prev = None
    for s_, e_, n_ in rows:
        if cls(n_, e_ - s_) != "trial_busy":
            continue
        ints = re.findall(r"(?<![\w])(\d+)(?=[,>])", n_)
        print(f"part {ints[-1] if ints else '?'}  S {ints[-2] if len(ints) > 1 else '?'}  {(e_ - s_) / 1e3:9.1f} us   gap {0 if prev is None else (s_ - prev) / 1e3:8.1f} us")
        prev = e_
if len(busy) <= skip + 1:
    raise SystemExit("no busy trial launches found")
lo, hi = busy[skip], busy[-1]
acc, prev_end = {}, None
for s, e, n in rows[lo:hi + 1]:
    c = cls(n, e - s)
    a = acc.setdefault(c, {"launches": 0, "dur_ns": 0, "gap_ns": 0})
    a["launches"] += 1
    a["dur_ns"] += e - s
    if prev_end is not None:
        a["gap_ns"] += max(0, s - prev_end)
    prev_end = max(prev_end or e, e)
span = rows[hi][1] - rows[lo][0]
passes = len(busy) - skip
out = {"busy_passes": passes, "span_ms": span / 1e6, "span_per_pass_us": span / passes / 1e3,
       "kernel_us_per_pass": {}, "gap_us_per_pass": {}, "classes": {}}
for c, a in sorted(acc.items()):
    out["classes"][c] = {"launches": a["launches"], "mean_us": a["dur_ns"] / a["launches"] / 1e3,
                         "mean_gap_before_us": a["gap_ns"] / a["launches"] / 1e3}
    out["kernel_us_per_pass"][c] = a["dur_ns"] / passes / 1e3
    out["gap_us_per_pass"][c] = a["gap_ns"] / passes / 1e3
# inside a solve: what lies between two consecutive busy passes with nothing but the solver's own small kernels
# (idle shape kernels, refresh-beta, set-max-iter) between them - no re-initialisation, no foreign kernels
inner, prev_busy_end, clean = [], None, True
for s_, e_, n_ in rows[lo:hi + 1]:
    c = cls(n_, e_ - s_)
    if c == "trial_busy":
        if prev_busy_end is not None and clean:
            inner.append((s_ - prev_busy_end) / 1e3)
        prev_busy_end, clean = e_, True
    elif c in ("other", "zf_init", "zf_eval_kernel"):
        clean = False
if inner:
    inner.sort()
    out["between_consecutive_busy_passes_us"] = {"pairs": len(inner), "median": inner[len(inner) // 2],
                                                 "p10": inner[len(inner) // 10], "p90": inner[len(inner) * 9 // 10]}
out["kernel_us_per_pass_total"] = sum(out["kernel_us_per_pass"].values())
out["gap_us_per_pass_total"] = sum(out["gap_us_per_pass"].values())
print(json.dumps(out, indent=1))
