#!/bin/bash
# the whole GPU suite, then the evidence sets of the two bench flag sets (kernel trace + PMC) with this round's build
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_suite
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5_suite/gpu_suite.log 2>&1; rc=$?
tail -6 gpurun_out/r5_suite/gpu_suite.log
[ $rc -ne 0 ] && exit $rc
bash tools/profile_r4.sh r05_driver_k20_w5 --steps 20 --warmup 5 > gpurun_out/r5_suite/prof_driver.log 2>&1; tail -2 gpurun_out/r5_suite/prof_driver.log
bash tools/profile_r4.sh r05_defaults_k100_w10 --steps 100 --warmup 10 > gpurun_out/r5_suite/prof_defaults.log 2>&1; tail -2 gpurun_out/r5_suite/prof_defaults.log
