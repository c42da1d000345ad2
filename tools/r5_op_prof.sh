#!/bin/bash
# per-kernel durations of the operator-form LASSO at 1024^2 and 4096^2 (rocprofv3 --kernel-trace --stats) -> gpurun_out/r5_op/
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_op
for sz in 1024 4096; do
  rm -rf /tmp/prof_$sz
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$sz -o op -- python3 tools/op_bench.py --size $sz --iters 100 > gpurun_out/r5_op/bench_${sz}_under_rocprof.json 2> /tmp/prof_$sz.err
  echo "rc $?"
  f=$(find /tmp/prof_$sz -type f -name "*kernel_stats*" | head -1)
  if [ -n "$f" ]; then cp $f gpurun_out/r5_op/kernel_stats_$sz.csv; head -10 $f | cut -c1-200; else find /tmp/prof_$sz -type f | head; tail -5 /tmp/prof_$sz.err; fi
done
exit 0
