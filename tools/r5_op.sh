#!/bin/bash
# operator LASSO: parity tests, then rate + per-kernel times at the notebook's size and beyond
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_op
timeout -k 10 600 python -m pytest tests/test_gpu_operator_lasso.py -x -q -m gpu > gpurun_out/r5_op/tests.log 2>&1; rc=$?
tail -15 gpurun_out/r5_op/tests.log
[ $rc -ne 0 ] && exit $rc
for sz in 256 1024 4096; do
  it=400; chk=3; [ $sz = 4096 ] && it=100 && chk=0
  python tools/op_bench.py --size $sz --iters $it --check $chk > gpurun_out/r5_op/bench_$sz.json 2> gpurun_out/r5_op/bench_$sz.err || { tail -5 gpurun_out/r5_op/bench_$sz.err; exit 1; }
  cat gpurun_out/r5_op/bench_$sz.json
  ZF_OP_SEPARABLE=0 python tools/op_bench.py --size $sz --iters $it > gpurun_out/r5_op/bench_${sz}_general.json 2>/dev/null && cat gpurun_out/r5_op/bench_${sz}_general.json
done
for sz in 1024 4096; do
  rm -rf /tmp/prof_$sz
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$sz -o op -- python3 tools/op_bench.py --size $sz --iters 100 > /dev/null 2>&1
  f=$(find /tmp/prof_$sz -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r5_op/kernel_stats_$sz.csv && head -8 $f
done
