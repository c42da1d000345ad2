#!/bin/bash
# The evidence behind bench.py's roofline block for ONE flag set, on the GPU box:   tools/profile_r4.sh <label> [bench flags...]
#   1. bench.py itself                                   -> gpurun_out/prof_<label>/bench.json
#   2. rocprofv3 --kernel-trace --stats                  -> .../trace/ (per-kernel durations) + bench_traced.json
#   3. rocprofv3 --pmc FETCH_SIZE, 4. --pmc WRITE_SIZE   (separate passes, no trace domains)
#   5. rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES
# then tools/pmc_summary_r4.py -> pmc.json (per shape-specific kernel; ZF_PROFILE_N = elements when not 1e8), tools/timeline.py -> timeline.json.  Copy what is to be judged into profiles/.
set -e -o pipefail
LABEL="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_$LABEL"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --no-regimes > "$OUT/bench_traced.json" 2> /dev/null
SHORT="--no-cpu-baseline --no-regimes --min-seconds 0.1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" "$@" $SHORT > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" "$@" $SHORT > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/clock" -o clock -- python3 "$ROOT/bench.py" "$@" $SHORT > /dev/null 2>&1 || echo "clock counters not collected"
python3 "$ROOT/tools/pmc_summary_r4.py" "$OUT" "${ZF_PROFILE_N:-100000000}" "$LABEL" > "$OUT/pmc.json"
python3 "$ROOT/tools/rocprof_busy_mean.py" "$OUT/trace" > "$OUT/kernel_trace_busy_mean.json"
python3 "$ROOT/tools/timeline.py" "$OUT/trace" 1 > "$OUT/timeline.json" || true
cp "$OUT"/trace/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || cp "$OUT"/trace/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || true
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/clock" "$OUT/trace"
ls "$OUT"
