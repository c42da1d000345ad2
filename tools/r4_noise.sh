#!/bin/bash
# A/B of builds / switches on the long solves that cross the resolution limit of the acceptance test (one box, alternating)
out=gpurun_out/r4g; mkdir -p $out
for rep in 1 2; do
  for v in base glds8 shortgen; do
    case $v in
      base) env="";;
      glds8) env="ZF_LIB_PATH=$PWD/_r2/v1/libzfista_hip.so";;
      shortgen) env="ZF_SHORT_VIA_GENERAL=1";;
    esac
    for n in 1e8 1e7; do
      k=300; [ $n = 1e7 ] && k=400
      echo "$v $n $(env $env python tools/long_run.py $n $k 2>/dev/null | tail -1)" >> $out/noise_ab.txt
    done
  done
done
cat $out/noise_ab.txt
