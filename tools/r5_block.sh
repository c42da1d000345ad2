#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_block
out=gpurun_out/r5_block/block_probe.jsonl; : > $out
for n in 30000 1000000 10000000; do
  for ev in "" "--events"; do
    for ra in 1 0; do
      echo -n "ra=$ra $ev " >> $out
      ZF_RUNAHEAD=$ra timeout -k 10 300 python tools/block_probe.py --n $n --steps 64 $ev >> $out 2>&1 || exit 1
    done
  done
done
cat $out
