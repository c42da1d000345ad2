#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_op
timeout -k 10 600 python -m pytest tests/test_gpu_operator_lasso.py -x -q -m gpu > gpurun_out/r5_op/tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5_op/tests.log
[ $rc -ne 0 ] && exit $rc
for sz in 256 1024 4096; do
  it=400; [ $sz = 4096 ] && it=100
  python tools/op_bench.py --size $sz --iters $it > gpurun_out/r5_op/bench_$sz.json 2> gpurun_out/r5_op/bench_$sz.err || { tail -5 gpurun_out/r5_op/bench_$sz.err; exit 1; }
  cat gpurun_out/r5_op/bench_$sz.json
done
bash tools/r5_op_prof.sh
