#!/bin/bash
# operator LASSO: workgroups that walk their tiles with the next tile's loads in flight, against a workgroup per tile (HEAD)
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_op
timeout -k 10 800 python -m pytest tests/test_gpu_operator_lasso.py -x -q -m gpu > gpurun_out/r5_op/tests_persist.log 2>&1; rc=$?
tail -3 gpurun_out/r5_op/tests_persist.log
[ $rc -ne 0 ] && exit $rc
out=gpurun_out/r5_op/persist_ab.txt; : > $out
for rep in 1 2; do for lib in head new new_one_tile_per_wg; do for sz in 256 1024 2048 4096; do
  unset ZF_LIB_PATH ZF_OP_PERSIST
  [ $lib = head ] && export ZF_LIB_PATH=$PWD/tools/bin/libzfista_hip_head.so
  [ $lib = new_one_tile_per_wg ] && export ZF_OP_PERSIST=0
  echo -n "$lib size $sz: " >> $out; python tools/op_bench.py --size $sz --iters 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['it_per_s'],1), 'it/s', round(d['ms_per_iteration'],4), 'ms')" >> $out
done; done; done
cat $out
