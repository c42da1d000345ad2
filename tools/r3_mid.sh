#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
python -m pytest tests/test_gpu_temporal.py tests/test_gpu_parity_diag.py tests/test_gpu_fuzz_parity.py tests/test_gpu_bench_contract.py tests/test_gpu_checkpoint.py -q -x 2>&1 | tail -3
for rep in 1 2 3; do
  for v in main NOMID; do
    unset ZF_LIB_PATH
    if [ "$v" != "main" ]; then export ZF_LIB_PATH="$R/zfista_amd/csrc/variants/libzf_$v.so"; fi
    python3 $R/bench.py --no-cpu-baseline --no-regimes --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v k20 %.0f' % d['value'], 'kernel %.4f' % d['roofline']['kernel_avg_ms'])"
  done
done
