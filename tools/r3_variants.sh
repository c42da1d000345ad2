#!/bin/bash
# A/B of kernel variants on the GPU box: tools/r3_variants.sh OUTDIR "A C main" -> one JSON line per (variant, workload)
OUT="$1"; shift
mkdir -p "$OUT"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for v in $1; do
  if [ "$v" = "main" ]; then unset ZF_LIB_PATH; else export ZF_LIB_PATH="$ROOT/zfista_amd/csrc/variants/libzf_$v.so"; fi
  for T in default 10; do
    if [ "$T" = "default" ]; then unset ZF_TILES_PER_WG; else export ZF_TILES_PER_WG=$T; fi
    python3 "$ROOT/tools/bench_configs.py" --cfg 2 --steps 100 --warmup 10 2>/dev/null | sed "s/^/{\"variant\": \"$v\", \"T\": \"$T\", \"r\": /; s/$/}/" >> "$OUT/variants.jsonl"
    python3 "$ROOT/tools/long_run.py" 1e7 400 2>/dev/null | sed "s/^/{\"variant\": \"$v\", \"T\": \"$T\", \"r\": /; s/$/}/" >> "$OUT/variants.jsonl"
  done
  unset ZF_TILES_PER_WG
  python3 "$ROOT/tools/long_run.py" 1e8 300 2>/dev/null | sed "s/^/{\"variant\": \"$v\", \"T\": \"default\", \"r\": /; s/$/}/" >> "$OUT/variants.jsonl"
  python3 "$ROOT/bench.py" --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | cut -c1-200 | sed "s/^/{\"variant\": \"$v\", \"bench\": \"k20\", \"r\": /; s/$/}/" >> "$OUT/variants_bench.txt"
  python3 "$ROOT/bench.py" --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'variant':'$v','bench':'k100','value':d['value'],'kernel_avg_ms':d['roofline']['kernel_avg_ms'],'other_ms':d['roofline']['other_passes_avg_ms'],'passes':d['config']['passes_per_block']}))" >> "$OUT/variants_bench.txt"
done
cat "$OUT/variants.jsonl" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['r']
    print(d['variant'], d['T'], r.get('workload', 'long n=%g' % r.get('n', 0)), 'it/s %.0f' % r['it_per_s'], 'kernel_ms', r.get('trial_kernel_ms', r.get('full_chain_ms')), 'other_ms', r.get('other_ms'), 'passes', r.get('passes'))
"
cat "$OUT/variants_bench.txt"
