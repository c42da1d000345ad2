#!/bin/bash
# acceptance="resolved" beside the reference's evaluation of the test: the bench's two regimes and the long solves across the noise floor
out=gpurun_out/r5_accept; mkdir -p $out; : > $out/ab.jsonl
b() { tag=$1; shift; line=$(python bench.py "$@" --no-cpu-baseline --no-regimes 2>/dev/null | tail -1); echo "{\"tag\": \"$tag\", \"line\": $line}" >> $out/ab.jsonl; echo "$tag done"; }
for rep in 1 2; do
  b k100_reference --steps 100 --warmup 10
  b k100_resolved --steps 100 --warmup 10 --acceptance resolved
  b k20_reference --steps 20 --warmup 5
  b k20_resolved --steps 20 --warmup 5 --acceptance resolved
  b k64_reference --steps 64 --warmup 16
  b k64_resolved --steps 64 --warmup 16 --acceptance resolved
  b cfg2_k100_reference --n 10000000 --steps 100 --warmup 10
  b cfg2_k100_resolved --n 10000000 --steps 100 --warmup 10 --acceptance resolved
  b cfg2_k64_reference --n 10000000 --steps 64 --warmup 16
  b cfg2_k64_resolved --n 10000000 --steps 64 --warmup 16 --acceptance resolved
done
for n in 1e8 1e7; do
  k=300; [ $n = 1e7 ] && k=400
  python tools/long_run.py $n $k 2>/dev/null | tail -1 > $out/long_${n}_reference.json
  python tools/long_run.py $n $k --acceptance resolved 2>/dev/null | tail -1 > $out/long_${n}_resolved.json
  python tools/long_run.py $n 1000 --acceptance resolved 2>/dev/null | tail -1 > $out/long_${n}_1000it_resolved.json
  python tools/long_run.py $n 1000 2>/dev/null | tail -1 > $out/long_${n}_1000it_reference.json
done
python - <<'PY'
import json, glob
for l in open("gpurun_out/r5_accept/ab.jsonl"):
    d = json.loads(l); b = d["line"]; r = b.get("roofline") or {}
    print(f"{d['tag']:24s} {b['value']:10.1f} it/s  passes/block {b['config'].get('passes_per_block')}  kernel_ms {r.get('kernel_avg_ms')}  frac {r.get('frac')}")
for f in sorted(glob.glob("gpurun_out/r5_accept/long_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unparsed"); continue
    print(f.split("/")[-1], {k: d.get(k) for k in ("iterations", "rejections", "lr_final", "passes", "it_per_s", "full_chain_ms", "other_ms")})
PY
