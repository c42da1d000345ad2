set -e
mkdir -p gpurun_out/coh
for rep in 1 2; do for v in c0 c1; do
  echo "== $v rep $rep"
  ZF_LIB_PATH=$PWD/_dbg/libzfista_$v.so timeout -k 10 200 python3 tools/bench_configs.py --cfg 2 --steps 64 --warmup 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2', round(d['it_per_s']), d['trial_kernel_ms'])"
  ZF_LIB_PATH=$PWD/_dbg/libzfista_$v.so timeout -k 10 300 python3 bench.py --steps 64 --warmup 16 --min-seconds 0.3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('n1e8', round(d['value']), d['roofline']['kernel_avg_ms'])"
done; done
