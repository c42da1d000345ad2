#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
for rep in 1 2 3; do
  for v in main S0N3 S1N2 r2; do
    unset ZF_LIB_PATH; B="$R/bench.py"; X="--no-regimes"
    if [ "$v" = "r2" ]; then B="$R/_r2/bench.py"; X=""; elif [ "$v" != "main" ]; then export ZF_LIB_PATH="$R/zfista_amd/csrc/variants/libzf_$v.so"; fi
    (cd $(dirname $B) && python3 $B --no-cpu-baseline $X --steps 20 --warmup 5 2>/dev/null) | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v k20 %.0f' % d['value'], 'kernel %.4f' % d['roofline']['kernel_avg_ms'])"
  done
done
