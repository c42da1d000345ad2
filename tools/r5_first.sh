#!/bin/bash
# round 5, first GPU call: the headline A/B (r3 tree vs HEAD) and the operator kernels beyond the notebook's size
export TMPDIR=/tmp
bash tools/r5_headline_ab.sh || exit 1
mkdir -p gpurun_out/r5_op
for sz in 1024 2048 4096; do
  it=50; [ $sz = 4096 ] && it=20
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5_op/$sz -o op -- python3 examples/deblur_operator_lasso.py --size $sz --iters $it --cpu-iters 2 > gpurun_out/r5_op/$sz.json 2> gpurun_out/r5_op/$sz.err || exit 1
  tail -1 gpurun_out/r5_op/$sz.json
done
