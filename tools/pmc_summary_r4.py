#!/usr/bin/env python3
"""Per-kernel summary of the rocprofv3 passes of tools/profile_r4.sh: for every shape-specific trial kernel that did
work (PART 0 = the full chain, PART 1 = the short-chain bodies, PART 2 = the general 16-trial body the driver's K = 20
blocks run) - HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate passes, corrected as MI355X_MICROARCH.md
prescribes for gfx950: FETCH_SIZE counts 16 B / lane streaming reads at half weight, both counters are KiB), duration,
engine clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), VALU instructions per launch and the busy fraction of the vector
pipes (rocprof's VALUBusy: SQ_ACTIVE_INST_VALU / 256 CUs / cycles).

    pmc_summary_r3.py <dir with fetch/ write/ clock/ [trace/]> <n> <label>"""
import collections
import csv
import glob
import json
import os
import re
import sys

out, n, label = sys.argv[1], int(float(sys.argv[2])), sys.argv[3]


def part_of(name):
    """zf_trial_kernel<GI, NEST, BOX, NT, S, HIST, PART, L>: "part<PART>", mid chains "part3_L<L>" (round 4: the
    template list ends with the mid-chain length)."""
    if "zf_runahead_kernel" in name:   # the run-ahead full chain (its duration includes the wait for the pass before it)
        return "runahead"
    ints = re.findall(r"(?<![\w])(\d+)(?=[,>])", name)
    if len(ints) < 3:
        return "part?"
    part, length = ints[-2], ints[-1]
    return f"part{part}_L{length}" if part == "3" else f"part{part}"


def rows(sub):
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if "zf_trial_kernel" in row.get("Kernel_Name", "") or "zf_runahead_kernel" in row.get("Kernel_Name", ""):
                    yield row


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "clock"):
    per_dispatch = collections.defaultdict(dict)
    for r in rows(sub):
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if dur < max(20_000, n // 1000):
            continue   # a launch that found another shape (or the solve finished) and exited
        key = (part_of(r["Kernel_Name"]), r["Dispatch_Id"])
        per_dispatch[key][r["Counter_Name"]] = float(r["Counter_Value"])
        per_dispatch[key]["_dur"] = dur
    for (part, _), d in per_dispatch.items():
        for c, v in d.items():
            acc[part][f"{sub}:{c}"].append(v)

res = {"label": label, "n": n, "model_bytes_per_launch": 48 * n,
       "corrections": "FETCH_SIZE x 1024 x 2 (gfx950 half-count of 16 B/lane streaming reads), WRITE_SIZE x 1024",
       "kernels": {}}
mean = lambda v: sum(v) / len(v) if v else None   # noqa: E731
med = lambda v: sorted(v)[len(v) // 2] if v else None   # noqa: E731
for part, d in sorted(acc.items()):
    k = {"busy_launches": {s: len(d.get(f"{s}:_dur", [])) for s in ("fetch", "write", "clock")}}
    f, w = d.get("fetch:FETCH_SIZE"), d.get("write:WRITE_SIZE")
    if f and w:
        k["read_bytes_per_launch_corrected"] = mean(f) * 1024 * 2
        k["write_bytes_per_launch"] = mean(w) * 1024
        k["hbm_bytes_per_launch"] = k["read_bytes_per_launch_corrected"] + k["write_bytes_per_launch"]
        k["ratio_traffic_over_model"] = k["hbm_bytes_per_launch"] / (48 * n)
    dur = d.get("clock:_dur")
    if dur:
        k["launch_ms_under_pmc"] = {"min": min(dur) / 1e6, "median": med(dur) / 1e6, "max": max(dur) / 1e6}
        gui = d.get("clock:GRBM_GUI_ACTIVE")
        if gui:
            clk = sorted(g / 8 / t for g, t in zip(gui, dur))
            k["engine_clock_GHz"] = {"min": clk[0], "median": clk[len(clk) // 2], "max": clk[-1]}
        valu = d.get("clock:SQ_INSTS_VALU")
        if valu:
            k["SQ_INSTS_VALU_per_launch"] = mean(valu)
            k["valu_lane_instructions_per_element"] = mean(valu) * 64 / n
        act, busy = d.get("clock:SQ_ACTIVE_INST_VALU"), d.get("clock:SQ_BUSY_CU_CYCLES")
        if act and gui:
            # rocprof's VALUBusy: SQ_ACTIVE_INST_VALU / CU_NUM / GRBM_GUI_ACTIVE per XCD (GUI_ACTIVE is summed over the 8 XCDs)
            k["valu_busy_fraction"] = mean([a / 256 / (g / 8) for a, g in zip(act, gui) if g > 0])
        if busy and gui:
            k["cu_busy_fraction"] = mean([b / 256 / (g / 8) for b, g in zip(busy, gui) if g > 0])
    res["kernels"][part] = k
print(json.dumps(res, indent=1))
