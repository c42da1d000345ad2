#!/bin/bash
# Where the cycles of the full-chain trial kernel go: two rocprofv3 --pmc passes (no trace domains)
# over a short bench run.    tools/pmc_valu_busy.sh [S]   ->  gpurun_out/valu_busy_sS/{a,b}/
set -e -o pipefail
S="${1:-16}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/valu_busy_s$S"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SHORT="--sub-iters $S --no-cpu-baseline --steps 64 --warmup 16 --min-seconds 0.1"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/a" -o a -- python3 "$ROOT/bench.py" $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/b" -o b -- python3 "$ROOT/bench.py" $SHORT > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "zf_trial_kernel" not in row["Kernel_Name"]:
            continue
        dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        if dur < 100_000:
            continue
        key = row["Kernel_Name"].split("(")[0]
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        acc[key]["_dur_ns_" + os.path.basename(os.path.dirname(path))[:1]].append(dur)
res = {}
for k, d in acc.items():
    res[k] = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()}
print(json.dumps(res, indent=1))
PY
