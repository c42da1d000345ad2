#!/bin/bash
# operator LASSO, this build against HEAD's library, by image size and blur size
export TMPDIR=/tmp
run() { # lib size extra
  unset ZF_LIB_PATH; [ $1 = head ] && export ZF_LIB_PATH=$PWD/tools/bin/libzfista_hip_head.so
  echo -n "$1 size $2 $3 $4 $5: "; python tools/op_bench.py --size $2 --iters 200 $3 $4 $5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['it_per_s'],1), 'it/s')"
}
timeout -k 10 800 python -m pytest tests/test_gpu_operator_lasso.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2; do for sz in 256 1024 2048 4096; do for lib in head new; do run $lib $sz; done; done; done
for sz in 2048 4096; do for k in 5 13 15; do for g in "" "--general"; do for lib in head new; do run $lib $sz --k $k $g; done; done; done; done
