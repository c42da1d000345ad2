#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_suite
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz_parity.py tests/test_gpu_bench_contract.py -x -q -m gpu > gpurun_out/r5_suite/fuzz_contract.log 2>&1; rc=$?
tail -5 gpurun_out/r5_suite/fuzz_contract.log
exit $rc
