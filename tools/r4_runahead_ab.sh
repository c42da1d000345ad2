#!/bin/bash
# cfg2 (P-diag n = 1e7) and the small sizes with run-ahead passes (default) and without (ZF_RUNAHEAD=0), one box,
# alternating:   tools/r4_runahead_ab.sh OUTFILE
OUT="$1"; : > "$OUT"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
B="python3 $ROOT/tools/bench_configs.py"
for rep in 1 2 3; do
  for ra in 1 0; do
    for kw in "64 16" "100 10" "400 10"; do
      set -- $kw
      ZF_RUNAHEAD=$ra $B --cfg 2 --steps $1 --warmup $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); d['runahead'] = $ra; d['K'] = $1; d['W'] = $2; print(json.dumps(d))" >> "$OUT"
    done
    ZF_RUNAHEAD=$ra python3 $ROOT/tools/long_run.py 1e7 400 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); d['runahead'] = $ra; d.pop('by_shape', None); print(json.dumps(d))" >> "$OUT"
  done
done
wc -l "$OUT"
