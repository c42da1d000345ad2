#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_op
timeout -k 10 900 python -m pytest tests/test_gpu_operator_lasso.py -x -q -m gpu > gpurun_out/r5_op/tests.log 2>&1; rc=$?
tail -8 gpurun_out/r5_op/tests.log
[ $rc -ne 0 ] && exit $rc
for sz in 256 1024 4096; do
  it=400; [ $sz = 4096 ] && it=200
  for ty in default 8 16 32; do
    if [ $ty = default ]; then e=""; else e="ZF_OP_TY=$ty"; fi
    echo -n "size $sz ty $ty: "; env $e python tools/op_bench.py --size $sz --iters $it 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['it_per_s'],1), 'it/s', round(d['ms_per_iteration'],4), 'ms', round(d['hbm_fraction_of_8TBps'],3))"
  done
done
python examples/deblur_sweep.py --check > gpurun_out/r5_op/sweep.json 2> gpurun_out/r5_op/sweep.err || tail -5 gpurun_out/r5_op/sweep.err
cat gpurun_out/r5_op/sweep.json
