#!/bin/bash
bash tools/profile_op_r5.sh 4096 60 > gpurun_out/r5_prof_op_4096.log 2>&1; tail -3 gpurun_out/r5_prof_op_4096.log
bash tools/profile_op_r5.sh 1024 200 > gpurun_out/r5_prof_op_1024.log 2>&1; tail -3 gpurun_out/r5_prof_op_1024.log
bash tools/r5_ahead_ab.sh
