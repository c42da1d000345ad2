#!/bin/bash
# same-box A/B of this tree against the round-2 tree (exported to _r2/ with its own build): bench.py at the driver's flags and at the defaults
R="${GRAFT_REPO_ROOT:-$PWD}"
for rep in 1 2 3; do
  for t in r3 r2; do
    if [ "$t" = "r3" ]; then B="$R/bench.py"; X="--no-regimes"; else B="$R/_r2/bench.py"; X=""; fi
    (cd $(dirname $B) && python3 $B --no-cpu-baseline $X --steps 20 --warmup 5 2>/dev/null) | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t k20 %.0f' % d['value'], 'kernel %.4f' % d['roofline']['kernel_avg_ms'])"
    (cd $(dirname $B) && python3 $B --no-cpu-baseline $X 2>/dev/null) | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$t k100 %.0f' % d['value'], 'full %.4f' % (r['full_chain_passes_avg_ms'] or 0), 'other %.4f' % (r['other_passes_avg_ms'] or 0), 'passes/block', d['config']['passes_per_block'])"
  done
done
