#!/bin/bash
# same-box A/B: in-kernel finalisation + predicted pass shape (default) vs the round-2 launch sequence
O="$1"; mkdir -p "$O"; R="${GRAFT_REPO_ROOT:-$PWD}"
for mode in new old new old; do
  if [ "$mode" = "old" ]; then export ZF_FIN_KERNEL=1 ZF_SPECULATE=0; else unset ZF_FIN_KERNEL ZF_SPECULATE; fi
  for args in "--cfg 2 --steps 64 --warmup 16" "--cfg 2 --steps 100 --warmup 10"; do
    python3 "$R/tools/bench_configs.py" $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', '$args', 'it/s %.0f' % d['it_per_s'], 'kernel_ms %.4f' % d['trial_kernel_ms'], 'passes', d['passes'], 'trials', d['trials'])"
  done
  python3 "$R/tools/long_run.py" 1e8 300 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode long1e8 it/s %.0f' % d['it_per_s'], 'full %.3f x%d' % (d['full_chain_ms'], d['full_chain_passes']), 'other %.3f x%d' % (d['other_ms'], d['other_passes']), 'rej', d['rejections'])"
  python3 "$R/bench.py" --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode bench k20 %.0f' % d['value'], 'kernel %.4f' % d['roofline']['kernel_avg_ms'])"
  python3 "$R/bench.py" --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$mode bench k100 %.0f' % d['value'], 'full %.4f' % (r['full_chain_passes_avg_ms'] or 0), 'other %.4f' % (r['other_passes_avg_ms'] or 0), 'passes/block', d['config']['passes_per_block'])"
done
