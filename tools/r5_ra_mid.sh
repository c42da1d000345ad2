#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_ra_mid
timeout -k 10 900 python -m pytest tests/test_gpu_runahead.py tests/test_gpu_accept.py tests/test_gpu_ahead.py -x -q -m gpu > gpurun_out/r5_ra_mid/tests.log 2>&1; rc=$?
tail -8 gpurun_out/r5_ra_mid/tests.log
[ $rc -ne 0 ] && exit $rc
out=gpurun_out/r5_ra_mid/ab.jsonl; : > $out
for rep in 1 2; do
 for ra in 0 1; do
  echo "ra=$ra n=1e7 K20 W5" >> $out
  ZF_RUNAHEAD=$ra timeout -k 10 300 python bench.py --n 10000000 --steps 20 --warmup 5 --no-cpu-baseline --no-regimes >> $out 2>/dev/null || exit 1
  echo "ra=$ra n=1e6 K64 W16" >> $out
  ZF_RUNAHEAD=$ra timeout -k 10 300 python tools/block_probe.py --n 1000000 --steps 64 >> $out 2>/dev/null || exit 1
  echo "ra=$ra n=1e6 K20" >> $out
  ZF_RUNAHEAD=$ra timeout -k 10 300 python tools/block_probe.py --n 1000000 --steps 20 >> $out 2>/dev/null || exit 1
 done
done
python - <<'PY'
import json
tag=None
for l in open('gpurun_out/r5_ra_mid/ab.jsonl'):
    l=l.strip()
    if l.startswith('ra='): tag=l
    elif l.startswith('{'):
        d=json.loads(l)
        print(tag, d.get('value', d.get('it_per_s')), d.get('ms_per_step', d.get('block_us')), d.get('config',{}).get('runahead') if 'config' in d else '')
PY
