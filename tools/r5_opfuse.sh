#!/bin/bash
# operator LASSO: the prox step fused into the adjoint kernel's epilogue (two launches per trial) against a launch of its own
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_op
timeout -k 10 600 python -m pytest tests/test_gpu_operator_lasso.py -x -q -m gpu > gpurun_out/r5_op/tests_fuse.log 2>&1; rc=$?
tail -3 gpurun_out/r5_op/tests_fuse.log
[ $rc -ne 0 ] && exit $rc
ZF_OP_FUSE_PROX=0 timeout -k 10 600 python -m pytest tests/test_gpu_operator_lasso.py -x -q -m gpu 2>&1 | tail -1
out=gpurun_out/r5_op/fuse_ab.txt; : > $out
for rep in 1 2; do for f in 0 1; do for sz in 256 1024 4096; do
  echo -n "fuse_prox $f size $sz: " >> $out; ZF_OP_FUSE_PROX=$f python tools/op_bench.py --size $sz --iters 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['it_per_s'],1), 'it/s', round(d['ms_per_iteration'],4), 'ms')" >> $out
done; done; done
cat $out
