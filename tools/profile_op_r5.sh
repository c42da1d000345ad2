#!/bin/bash
# Counters of the operator-form LASSO kernels (zf_op_apply_kernel, zf_op_adjoint_kernel, the prox step) at one image size:
#   tools/profile_op_r5.sh <size> [iters]
#   1. rocprofv3 --kernel-trace --stats   2. --pmc FETCH_SIZE   3. --pmc WRITE_SIZE (separate passes)
#   4. --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE
# -> gpurun_out/prof_op_<size>/{kernel_stats.csv, pmc.json}.  Byte counters are calibrated IN THE SAME RUN on two kernels of known
# traffic with the same 8-byte-per-lane access width (the guide calibrates FETCH_SIZE for 16-byte-per-lane streams only):
# zf_eval_kernel<false,false> reads n doubles once, zf_resid_x_wide_kernel reads 2 n doubles.
set -e -o pipefail
SZ="$1"; IT="${2:-100}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_op_$SZ"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/op_bench.py --size $SZ --iters $IT"
$CMD > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/bench_traced.json" 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/lds" -o lds -- $CMD > /dev/null 2>&1 || echo "lds counters not collected"
python3 "$ROOT/tools/pmc_summary_op_r5.py" "$OUT" "$SZ" > "$OUT/pmc.json"
cp "$OUT"/trace/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || cp "$OUT"/trace/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || true
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/lds" "$OUT/trace"
cat "$OUT/pmc.json"
