#!/bin/bash
# Collect the evidence bench.py's roofline block cites, on the GPU box, for one chain length S
# (default 8; S=1 is the one-iteration-per-pass kernel):      tools/profile_bench.sh [S]
#   1. bench.py itself                      -> gpurun_out/prof_sS/bench.json
#   2. rocprofv3 --kernel-trace --stats     -> gpurun_out/prof_sS/trace/   (per-kernel average durations)
#   3. rocprofv3 --pmc FETCH_SIZE           -> gpurun_out/prof_sS/fetch/   (separate pass, no trace domains)
#   4. rocprofv3 --pmc WRITE_SIZE           -> gpurun_out/prof_sS/write/
#   5. rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES -> gpurun_out/prof_sS/clock/
# then tools/pmc_summary.py turns 3 + 4 (+ 5) into the bytes-per-launch JSON.  Copy the summaries you
# want judged into profiles/ (gpurun_out/ is scratch).
set -e -o pipefail
S="${1:-8}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_s$S"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
NOCPU="--no-cpu-baseline"
[ "$S" = "8" ] && NOCPU=""
python3 "$ROOT/bench.py" --sub-iters "$S" $NOCPU > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --sub-iters "$S" --no-cpu-baseline > "$OUT/bench_traced.json" 2> /dev/null
SHORT="--sub-iters $S --no-cpu-baseline --steps 40 --warmup 8 --min-seconds 0.1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" $SHORT > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" $SHORT > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/clock" -o clock -- python3 "$ROOT/bench.py" --sub-iters "$S" --no-cpu-baseline > /dev/null 2>&1 || echo "clock counters not collected"
python3 "$ROOT/tools/pmc_summary.py" "$OUT" 100000000 "$S" > "$OUT/pmc_traffic.json"
ls "$OUT" "$OUT/trace" | head -30
