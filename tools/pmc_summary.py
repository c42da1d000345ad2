#!/usr/bin/env python3
"""HBM bytes per launch of the fused trial kernel from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; tools/profile_bench.sh), corrected as MI355X_MICROARCH.md
prescribes for gfx950: FETCH_SIZE counts 16 B/lane streaming reads at half weight (x 2),
both counters are in KiB.  With a third pass (GRBM_GUI_ACTIVE, SQ_INSTS_VALU) also the
effective engine clock and the VALU instructions per element and trial.

    pmc_summary.py <dir with fetch/ write/ [clock/] [trace/]> [n] [S]"""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 8


def counter_rows(sub, counter, with_time=False):
    vals = []
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and "zf_trial_kernel" in row.get("Kernel_Name", ""):
                    if with_time:
                        vals.append((float(row["Counter_Value"]), int(row["Start_Timestamp"]), int(row["End_Timestamp"]),
                                     row.get("Dispatch_Id")))
                    else:
                        vals.append(float(row["Counter_Value"]))
    return vals


fetch = counter_rows("fetch", "FETCH_SIZE")
write = counter_rows("write", "WRITE_SIZE")
# full-chain passes only: launches enqueued after the solve has stopped exit at once (no traffic),
# and the last pass before max_iter may run a shorter chain (one iterate written instead of two)
full_w = [w for w in write if w > 0.9 * max(write)] if write else []
fetch = [f for f in fetch if f > 0.9 * max(fetch)] if fetch else []
model = (48 if S > 1 else 40) * n
res = {
    "command": f"rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --sub-iters {S} --no-cpu-baseline "
               "--steps 40 --warmup 8 --min-seconds 0.1  (one run per counter, no trace domains; tools/profile_bench.sh)",
    "kernel": f"zf_trial_kernel<grad inline, nesterov, nt, S={S}> (P-diag)",
    "n": n,
    "sub_iters": S,
    "FETCH_SIZE_raw_KiB_mean": sum(fetch) / max(len(fetch), 1),
    "FETCH_SIZE_full_chain_launches": len(fetch),
    "WRITE_SIZE_raw_KiB_mean_full_chain": sum(full_w) / max(len(full_w), 1),
    "WRITE_SIZE_launches": len(write),
    "WRITE_SIZE_full_chain_launches": len(full_w),
    "corrections": "FETCH_SIZE x 1024 x 2 (gfx950 half-count of 16 B/lane streaming reads, MI355X_MICROARCH.md "
                   "HBM); WRITE_SIZE x 1024",
}
res["read_bytes_per_launch_corrected"] = res["FETCH_SIZE_raw_KiB_mean"] * 1024 * 2
res["write_bytes_per_launch"] = res["WRITE_SIZE_raw_KiB_mean_full_chain"] * 1024
res["hbm_bytes_per_launch"] = res["read_bytes_per_launch_corrected"] + res["write_bytes_per_launch"]
res["model_bytes_per_launch"] = model
res["ratio_traffic_over_model"] = res["hbm_bytes_per_launch"] / model
res["algorithmic_bytes_per_launch"] = 40 * n * S

# optional: engine clock and VALU instruction count per launch.  Full-chain passes all execute the
# same instruction stream: they are the most frequent SQ_INSTS_VALU value (replay passes issue more,
# the last pass before max_iter less).  GRBM_GUI_ACTIVE is summed over the 8 XCDs.
gui = counter_rows("clock", "GRBM_GUI_ACTIVE", with_time=True)
valu = counter_rows("clock", "SQ_INSTS_VALU", with_time=True)
if gui and valu:
    import collections

    # (a pass of a 16-chain solver is two launches, one of which exits at once: the launches that did
    #  no work - also those enqueued after the solve stopped - are left out by their duration)
    busy = [v for v in valu if v[2] - v[1] > 100_000]
    mode = collections.Counter(round(v[0] / 1e6) for v in busy).most_common(1)[0][0]
    full_ids = {v[3] for v in busy if round(v[0] / 1e6) == mode}
    full_valu = [v[0] for v in valu if v[3] in full_ids]
    clocks = sorted(cyc / 8 / (t1 - t0) for cyc, t0, t1, did in gui if did in full_ids and t1 > t0)
    durs = sorted((t1 - t0) / 1e6 for cyc, t0, t1, did in gui if did in full_ids and t1 > t0)
    if clocks:
        res["engine_clock_GHz"] = {"min": clocks[0], "median": clocks[len(clocks) // 2], "max": clocks[-1],
                                   "launches": len(clocks),
                                   "how": "GRBM_GUI_ACTIVE / 8 XCDs / (End_Timestamp - Start_Timestamp) per full-chain launch"}
        res["full_chain_launch_ms_under_pmc"] = {"min": durs[0], "median": durs[len(durs) // 2], "max": durs[-1]}
        res["SQ_INSTS_VALU_per_launch"] = sum(full_valu) / len(full_valu)
        res["valu_instructions_per_element_trial_measured"] = sum(full_valu) / len(full_valu) * 64 / (n * S)
        res["valu_issue_ms_at_median_clock"] = (sum(full_valu) / len(full_valu) * 64 / (256 * 4 * 16)
                                                / (clocks[len(clocks) // 2] * 1e9) * 1e3)
print(json.dumps(res, indent=1))
