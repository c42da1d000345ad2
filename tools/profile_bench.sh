#!/bin/bash
# Collect the evidence bench.py's roofline block cites, on the GPU box:
#   1. bench.py itself                      -> gpurun_out/prof/bench.json
#   2. rocprofv3 --kernel-trace --stats     -> gpurun_out/prof/trace/   (per-kernel average durations)
#   3. rocprofv3 --pmc FETCH_SIZE           -> gpurun_out/prof/fetch/   (separate pass, no trace domains)
#   4. rocprofv3 --pmc WRITE_SIZE           -> gpurun_out/prof/write/
# then tools/pmc_summary.py turns 3 + 4 into the bytes-per-launch JSON.  Copy the summaries you
# want judged into profiles/ (gpurun_out/ is scratch).
set -e -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
# (tiles per workgroup pinned to what the autotuner picks on this device class, so that the trace
#  holds passes of the solve only, not the autotuner's probe launches)
ZF_TILES_PER_WG=8 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_traced.json" 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 40 --warmup 8 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 40 --warmup 8 > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/pmc_traffic.json"
ls -R "$OUT" | head -40
