#!/bin/bash
# cfg4 device search: totals polled directly (one dependent trip fewer per hand-over) against the flag-then-totals protocol
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_mo
timeout -k 10 900 python -m pytest tests/test_gpu_multiobjective.py tests/test_gpu_problem_library.py tests/test_gpu_mo_fullsize.py -x -q -m gpu > gpurun_out/r5_mo/tests_combine.log 2>&1; rc=$?
tail -4 gpurun_out/r5_mo/tests_combine.log
[ $rc -ne 0 ] && exit $rc
out=gpurun_out/r5_mo/cfg4_combine_ab.jsonl; : > $out
for rep in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export ZF_LIB_PATH=$PWD/tools/bin/libzfista_hip_old_mo.so; else unset ZF_LIB_PATH; fi
    echo -n "$lib " >> $out
    timeout -k 10 300 python tools/bench_configs.py --cfg 4 --steps 300 --dual-solver device 2>/dev/null | tail -1 >> $out || exit 1
  done
done
cat $out
