#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5b
timeout -k 10 900 python -m pytest tests/test_gpu_ahead.py tests/test_gpu_accept.py -x -q -m gpu -k "not communicator_equal and not several_rounds" > gpurun_out/r5b/new_tests.log 2>&1; rc=$?
tail -40 gpurun_out/r5b/new_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_runahead.py tests/test_gpu_libcomm.py tests/test_gpu_temporal.py tests/test_gpu_parity_diag.py tests/test_gpu_noise_floor.py tests/test_gpu_cfg5_fullsize.py tests/test_gpu_checkpoint.py -x -q -m gpu > gpurun_out/r5b/old_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r5b/old_tests.log
exit $rc
