#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do for b in 0 1; do for sz in 1024 4096; do
  echo -n "bands $b size $sz: "; ZF_OP_XCD_BANDS=$b python tools/op_bench.py --size $sz --iters 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['it_per_s'],1), 'it/s', round(d['ms_per_iteration'],4), 'ms')"
done; done; done
timeout -k 10 600 python -m pytest tests/test_gpu_operator_lasso.py -x -q -m gpu 2>&1 | tail -2
