#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_suite
timeout -k 10 1000 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_checkpoint.py tests/test_gpu_libcomm.py tests/test_gpu_multiprocess.py tests/test_gpu_harness.py tests/test_gpu_integration_doc.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r5_suite/gpu_suite_rest.log 2>&1; rc=$?
tail -6 gpurun_out/r5_suite/gpu_suite_rest.log
[ $rc -ne 0 ] && exit $rc
bash tools/profile_r4.sh r05_driver_k20_w5 --steps 20 --warmup 5 > gpurun_out/r5_suite/prof_driver.log 2>&1; tail -2 gpurun_out/r5_suite/prof_driver.log
bash tools/profile_r4.sh r05_defaults_k100_w10 --steps 100 --warmup 10 > gpurun_out/r5_suite/prof_defaults.log 2>&1; tail -2 gpurun_out/r5_suite/prof_defaults.log
