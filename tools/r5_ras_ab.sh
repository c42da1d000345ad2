#!/bin/bash
# sharded run-ahead passes (one-round grids behind the library communicator, world 1 = the 1-rank RCCL communicator) against
# passes ahead (ZF_RUNAHEAD_SHARDED=0), one pass at a time (ZF_AHEAD=0 too) and the unsharded solve
export TMPDIR=/tmp
mkdir -p gpurun_out/ras
out=gpurun_out/ras/ab.jsonl; : > $out
line() { python bench.py "$@" --no-cpu-baseline --no-regimes 2>/dev/null | tail -1; }
for rep in 1 2; do
  for n in 10000000 1000000; do
    for K in "64 16" "20 5"; do
      set -- $K
      echo "unsharded n=$n K=$1 W=$2" >> $out; line --n $n --steps $1 --warmup $2 >> $out
      echo "libcomm run-ahead n=$n K=$1 W=$2" >> $out; line --libcomm --n $n --steps $1 --warmup $2 >> $out
      echo "libcomm passes-ahead n=$n K=$1 W=$2" >> $out; ZF_RUNAHEAD_SHARDED=0 line --libcomm --n $n --steps $1 --warmup $2 >> $out
      echo "libcomm sequential n=$n K=$1 W=$2" >> $out; ZF_RUNAHEAD_SHARDED=0 ZF_AHEAD=0 line --libcomm --n $n --steps $1 --warmup $2 >> $out
    done
  done
done
python - <<'PY'
import json
tag=None
for l in open('gpurun_out/ras/ab.jsonl'):
    l=l.strip()
    if not l: continue
    if l.startswith('{'):
        d=json.loads(l); n=d['config'].get('n_per_gpu') or d['config'].get('n')
        print(tag.ljust(44), round(d['value']*1e8/float(tag.split('n=')[1].split()[0]),0) if False else d['value'], d['ms_per_step'], d['config'].get('runahead',{}).get('passes'), d['config'].get('passes_ahead'))
    else: tag=l
PY
