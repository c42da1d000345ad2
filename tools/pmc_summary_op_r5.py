#!/usr/bin/env python3
"""Per-kernel summary of tools/profile_op_r5.sh: HBM bytes per launch of the operator kernels from FETCH_SIZE / WRITE_SIZE
(KiB, separate passes), CALIBRATED on two kernels of the same run with known traffic and the same 8-byte-per-lane access
width (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count in your own access
pattern"), LDS instructions / bank-conflict cycles / busy cycles, VALU instructions, engine clock.
    pmc_summary_op_r5.py <dir with fetch/ write/ lds/> <image size>"""
import collections
import csv
import glob
import json
import os
import sys

out, size = sys.argv[1], int(sys.argv[2])
n = size * size
KERNELS = {"zf_op_apply_kernel": "apply (B W^-1 x+, fused: share of |s+ - b|^2, last workgroup decides)",
           "zf_op_adjoint_kernel": "adjoint (W B r, fused: residual at y formed in the tile load)",
           "zf_trial_kernel": "prox step (gradient vector in HBM, one trial)",
           "zf_eval_kernel": "calibration: reads n doubles (8 B per lane), writes nothing to speak of",
           "zf_resid_x_wide_kernel": "calibration: reads 2 n doubles (8 B per lane)"}
# algorithmic bytes per launch (no halo re-reads): what a perfect kernel moves
MODEL = {"zf_op_apply_kernel": (8 + 8) * n + 8 * n, "zf_op_adjoint_kernel": 24 * n + 8 * n, "zf_trial_kernel": 24 * n + 8 * n,
         "zf_eval_kernel": 8 * n, "zf_resid_x_wide_kernel": 16 * n}
MODEL_READ = {"zf_op_apply_kernel": 16 * n, "zf_op_adjoint_kernel": 24 * n, "zf_trial_kernel": 24 * n, "zf_eval_kernel": 8 * n,
              "zf_resid_x_wide_kernel": 16 * n}


def which(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "lds"):
    per = collections.defaultdict(dict)
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                k = which(r.get("Kernel_Name", ""))
                if not k:
                    continue
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                key = (k, r["Dispatch_Id"])
                per[key][r["Counter_Name"]] = float(r["Counter_Value"])
                per[key]["_dur"] = dur
    for (k, _), d in per.items():
        # launches that found the solve finished and left at once are not the kernel (the median duration tells them apart)
        acc[k][f"{sub}:_all"].append(d)
# (round 5: up to 5 Mi pixels the prox step rides in the adjoint kernel - no separate zf_trial_kernel in the loop: that kernel
#  then also reads x_k, x_{k-1} and writes x+ in place of the gradient)
fused = not any("fetch:_all" in d and len(d["fetch:_all"]) > 8 for k, d in acc.items() if k == "zf_trial_kernel")
if fused:
    KERNELS["zf_op_adjoint_kernel"] = "adjoint + prox step (W B r with the residual at y formed in the tile load; x+ = prox(y - lr grad) in the epilogue)"
    MODEL["zf_op_adjoint_kernel"] = (24 + 16) * n + 8 * n
    MODEL_READ["zf_op_adjoint_kernel"] = (24 + 16) * n
mean = lambda v: sum(v) / len(v) if v else None   # noqa: E731
res = {"image": f"{size} x {size}", "n": n, "prox_step_fused_into_adjoint": fused, "kernels": {}, "units": "FETCH_SIZE / WRITE_SIZE in KiB; bytes = counter x 1024"}
raw = {}
for k, d in acc.items():
    e = {"what": KERNELS[k], "model_bytes_per_launch": MODEL[k]}
    for sub in ("fetch", "write", "lds"):
        rows = d.get(f"{sub}:_all", [])
        if not rows:
            continue
        durs = sorted(r["_dur"] for r in rows)
        med = durs[len(durs) // 2]
        busy = [r for r in rows if r["_dur"] >= 0.5 * med]
        e.setdefault("busy_launches", {})[sub] = len(busy)
        e.setdefault("launch_us_under_pmc", {})[sub] = mean([r["_dur"] for r in busy]) / 1e3
        for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"):
            vals = [r[c] for r in busy if c in r]
            if vals:
                e[c + "_per_launch"] = mean(vals)
        if sub == "lds" and "GRBM_GUI_ACTIVE_per_launch" in e:
            e["engine_clock_GHz"] = e["GRBM_GUI_ACTIVE_per_launch"] / 8 / (e["launch_us_under_pmc"]["lds"] * 1e3)
    raw[k] = e
# calibration of the read counter for 8-byte-per-lane streams
cal = {}
for k in ("zf_eval_kernel", "zf_resid_x_wide_kernel"):
    if k in raw and "FETCH_SIZE_per_launch" in raw[k]:
        cal[k] = MODEL_READ[k] / (raw[k]["FETCH_SIZE_per_launch"] * 1024)
factor = mean(list(cal.values())) if cal else None
res["read_calibration"] = {"true bytes / (FETCH_SIZE x 1024) on kernels of known traffic": cal, "factor_used": factor}
for k, e in raw.items():
    if "FETCH_SIZE_per_launch" in e and factor:
        e["read_bytes_per_launch_calibrated"] = e["FETCH_SIZE_per_launch"] * 1024 * factor
    if "WRITE_SIZE_per_launch" in e:
        e["write_bytes_per_launch"] = e["WRITE_SIZE_per_launch"] * 1024
    if "read_bytes_per_launch_calibrated" in e and "write_bytes_per_launch" in e:
        e["hbm_bytes_per_launch"] = e["read_bytes_per_launch_calibrated"] + e["write_bytes_per_launch"]
        e["ratio_traffic_over_model"] = e["hbm_bytes_per_launch"] / e["model_bytes_per_launch"]
        us = e["launch_us_under_pmc"].get("fetch")
        if us:
            e["hbm_TBps_moved"] = e["hbm_bytes_per_launch"] / (us * 1e-6) / 1e12
            e["fraction_of_8TBps_moved"] = e["hbm_TBps_moved"] / 8.0
            e["fraction_of_8TBps_algorithmic"] = e["model_bytes_per_launch"] / (us * 1e-6) / 8e12
    if "SQ_INSTS_LDS_per_launch" in e:
        e["lds_instructions_per_pixel"] = e["SQ_INSTS_LDS_per_launch"] * 64 / n
        if e.get("SQ_LDS_IDX_ACTIVE_per_launch"):
            e["lds_bank_conflict_share_of_lds_cycles"] = e.get("SQ_LDS_BANK_CONFLICT_per_launch", 0.0) / e["SQ_LDS_IDX_ACTIVE_per_launch"]
    if "SQ_INSTS_VALU_per_launch" in e:
        e["valu_lane_instructions_per_pixel"] = e["SQ_INSTS_VALU_per_launch"] * 64 / n
    res["kernels"][k] = e
print(json.dumps(res, indent=1))
