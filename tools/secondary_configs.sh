#!/bin/bash
# One JSON line per secondary configuration (BASELINE cfg1-cfg4, the long solves, n = 8e8 on one GPU) with the
# current build, on the GPU box:   tools/secondary_configs.sh OUTFILE
OUT="$1"; : > "$OUT"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
B="python3 $ROOT/tools/bench_configs.py"
$B --cfg 1 --steps 100 --warmup 10 2>/dev/null >> "$OUT"
$B --cfg 2 --steps 100 --warmup 10 2>/dev/null >> "$OUT"
$B --cfg 3 --steps 100 --warmup 10 2>/dev/null >> "$OUT"
$B --cfg 2 --steps 64 --warmup 16 2>/dev/null >> "$OUT"
$B --cfg 2 --steps 400 --warmup 10 2>/dev/null >> "$OUT"
$B --cfg 4 --steps 100 --warmup 5 --dual-solver device 2>/dev/null >> "$OUT"
$B --cfg 4 --steps 100 --warmup 5 --dual-solver native 2>/dev/null >> "$OUT"
$B --cfg 4 --steps 10 --warmup 2 --dual-solver scipy 2>/dev/null >> "$OUT"
python3 "$ROOT/tools/long_run.py" 1e8 300 2>/dev/null >> "$OUT"
python3 "$ROOT/tools/long_run.py" 1e7 400 2>/dev/null >> "$OUT"
python3 "$ROOT/bench.py" --total-n 800000000 --no-cpu-baseline --no-regimes 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'workload': 'P-diag n = 8e8 on ONE GPU (bench.py --total-n 800000000, defaults K = 100, W = 10)', 'value_it_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> "$OUT"
wc -l "$OUT"
