#!/bin/bash
# same-box A/B of the DMA pipeline depth (ZF_GLDS_STAGES = 2 / 3 / 4): tools/r3_stages.sh OUTDIR
O="$1"; mkdir -p "$O"; R="${GRAFT_REPO_ROOT:-$PWD}"
for rep in 1 2; do
for v in main N2; do
  if [ "$v" = "main" ]; then unset ZF_LIB_PATH; else export ZF_LIB_PATH="$R/zfista_amd/csrc/variants/libzf_$v.so"; fi
  python3 "$R/bench.py" --no-cpu-baseline --no-regimes --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'variant':'$v','bench':'k20','value':d['value'],'kernel_ms':d['roofline']['kernel_avg_ms']}))" | tee -a "$O/stages.jsonl"
  python3 "$R/bench.py" --no-cpu-baseline --no-regimes 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(json.dumps({'variant':'$v','bench':'k100','value':d['value'],'full_ms':r['full_chain_passes_avg_ms'],'other_ms':r['other_passes_avg_ms']}))" | tee -a "$O/stages.jsonl"
  python3 "$R/tools/bench_configs.py" --cfg 2 --steps 64 --warmup 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'variant':'$v','bench':'cfg2_k64','value':d['it_per_s'],'kernel_ms':d['trial_kernel_ms']}))" | tee -a "$O/stages.jsonl"
done; done
