#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_mo
timeout -k 10 900 python -m pytest tests/test_gpu_libcomm.py tests/test_gpu_multiobjective.py tests/test_gpu_problem_library.py -x -q -m gpu > gpurun_out/r5_mo/tests.log 2>&1; rc=$?
tail -8 gpurun_out/r5_mo/tests.log
[ $rc -ne 0 ] && exit $rc
python tools/mo_sharded_rate.py 200 > gpurun_out/r5_mo/rate.json 2> gpurun_out/r5_mo/rate.err || tail -5 gpurun_out/r5_mo/rate.err
cat gpurun_out/r5_mo/rate.json
