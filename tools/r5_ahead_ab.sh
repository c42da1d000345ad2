#!/bin/bash
# Same-box A/B of passes ahead at kernel granularity: the sharded step sequence through a 1-rank RCCL communicator
# (bench.py --libcomm) with ZF_AHEAD=0 / 1 beside the unsharded solve, and the unsharded scheme (ZF_AHEAD_UNSHARDED=0 / 1)
# on the headline size and cfg2.  One JSON line per run -> gpurun_out/r5_ahead/ab.jsonl, a table on stdout.
out=gpurun_out/r5_ahead; mkdir -p $out; : > $out/ab.jsonl
run() {  # tag, env, bench args
  tag=$1; envs=$2; shift 2
  line=$(env $envs python bench.py "$@" --no-cpu-baseline --no-regimes 2>/dev/null | tail -1)
  echo "{\"tag\": \"$tag\", \"env\": \"$envs\", \"args\": \"$*\", \"line\": $line}" >> $out/ab.jsonl
  echo "$tag done"
}
for rep in 1 2; do
  run libcomm_n1e7_k64_ahead0 "ZF_AHEAD=0" --libcomm --n 10000000 --steps 64 --warmup 16
  run libcomm_n1e7_k64_ahead1 "ZF_AHEAD=1" --libcomm --n 10000000 --steps 64 --warmup 16
  run unsharded_n1e7_k64 "ZF_X=0" --n 10000000 --steps 64 --warmup 16
  run unsharded_n1e7_k64_ra0 "ZF_RUNAHEAD=0" --n 10000000 --steps 64 --warmup 16
  run unsharded_n1e7_k64_ahead_only "ZF_RUNAHEAD=0 ZF_AHEAD_UNSHARDED=1" --n 10000000 --steps 64 --warmup 16
  run unsharded_n1e7_k20_au0 "ZF_AHEAD_UNSHARDED=0" --n 10000000 --steps 20 --warmup 5
  run unsharded_n1e7_k20_au1 "ZF_AHEAD_UNSHARDED=1" --n 10000000 --steps 20 --warmup 5
  run libcomm_n1e8_k64_ahead0 "ZF_AHEAD=0" --libcomm --steps 64 --warmup 16
  run libcomm_n1e8_k64_ahead1 "ZF_AHEAD=1" --libcomm --steps 64 --warmup 16
  run unsharded_n1e8_k64_au0 "ZF_AHEAD_UNSHARDED=0" --steps 64 --warmup 16
  run unsharded_n1e8_k64_au1 "ZF_AHEAD_UNSHARDED=1" --steps 64 --warmup 16
  run libcomm_n1e8_k20_ahead0 "ZF_AHEAD=0" --libcomm --steps 20 --warmup 5
  run libcomm_n1e8_k20_ahead1 "ZF_AHEAD=1" --libcomm --steps 20 --warmup 5
  run unsharded_n1e8_k20_au0 "ZF_AHEAD_UNSHARDED=0" --steps 20 --warmup 5
  run unsharded_n1e8_k20_au1 "ZF_AHEAD_UNSHARDED=1" --steps 20 --warmup 5
done
python - <<'PY'
import json
for l in open("gpurun_out/r5_ahead/ab.jsonl"):
    d = json.loads(l); b = d["line"]; c = b["config"]
    print(f"{d['tag']:34s} {b['value']:10.1f} it/s  ms/step {b['ms_per_step']:.5f}  passes/block {c.get('passes_per_block')}  ahead {c.get('passes_ahead')}  ra {c.get('runahead', {}).get('launched_behind_a_pass_in_flight')}  exch_ms {(c.get('rccl') or {}).get('exchange_ms_per_pass')}")
PY
