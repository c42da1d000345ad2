#!/bin/bash
# cfg3 (dense LASSO 16384 x 65536): matrix-pipe utilisation of the column sweep and kernel durations with the current
# build, on the GPU box:   tools/cfg3_mfma_pmc.sh OUTDIR     (counters in their own pass, no trace domains)
set -e -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$(mkdir -p "$1" && cd "$1" && pwd)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc" -o pmc -- python3 "$ROOT/tools/bench_configs.py" --cfg 3 --steps 5 --warmup 1 > "$OUT/bench_under_pmc.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/tools/bench_configs.py" --cfg 3 --steps 50 --warmup 5 > "$OUT/bench_traced.json" 2>/dev/null
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path, newline="")):
        name = r["Kernel_Name"].split("(")[0]
        if "gemv" not in name:
            continue
        per[(name, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for (name, _), d in per.items():
        for c, v in d.items():
            acc[name][c].append(v)
res = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -- python3 tools/bench_configs.py --cfg 3 --steps 5 --warmup 1",
       "matrix": "16384 x 65536 fp64 (8 GiB), one sweep = 8.59 GB", "kernels": {}}
for name, d in acc.items():
    k = {c: sum(v) / len(v) for c, v in d.items()}
    k["launches"] = len(next(iter(d.values())))
    if k.get("SQ_WAVE_CYCLES"):
        # busy cycles of the matrix pipes per SIMD cycle: GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
        k["mfma_busy_fraction_of_simd_cycles"] = k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (k["GRBM_GUI_ACTIVE"] / 8 * 256 * 4) if k.get("GRBM_GUI_ACTIVE") else None
    res["kernels"][name] = k
stats = {}
for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        if "gemv" in r["Name"] or "trial" in r["Name"] or "resid" in r["Name"] or "finalize" in r["Name"]:
            stats[r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
res["kernel_stats_50_iterations"] = stats
res["bench"] = json.loads(open(os.path.join(out, "bench_traced.json")).read().strip().splitlines()[-1])
print(json.dumps(res, indent=1))
PY
