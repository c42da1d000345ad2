#!/bin/bash
# same-box A/B of the default library against zfista_amd/csrc/variants/libzf_$1.so on the noise-regime workloads
#   tools/ab_variant.sh VARIANT [reps]
V="$1"; REPS="${2:-2}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for rep in $(seq $REPS); do
for v in main $V; do
  if [ "$v" = "main" ]; then unset ZF_LIB_PATH; else export ZF_LIB_PATH="$ROOT/zfista_amd/csrc/variants/libzf_$v.so"; fi
  python3 "$ROOT/tools/long_run.py" 1e8 300 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'long 1e8/300', round(d['it_per_s']), 'full_ms', round(d['full_chain_ms'],4), 'other_ms', round(d['other_ms'],4), d['passes'], d['rejections'])"
  python3 "$ROOT/tools/long_run.py" 1e7 400 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'long 1e7/400', round(d['it_per_s']), 'full_ms', round(d['full_chain_ms'],4), 'other_ms', round(d['other_ms'],4), d['passes'], d['rejections'])"
  python3 "$ROOT/bench.py" --no-cpu-baseline --no-regimes 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'bench k100', round(d['value']), d['roofline'].get('other_passes_avg_ms'))"
done
done
