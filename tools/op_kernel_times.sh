#!/bin/bash
# busy durations of the operator kernels (median over launches that ran a trial) for one or two builds of the library
#   tools/op_kernel_times.sh <size> [lib.so ...]   ("-" = the in-tree build)
SZ=$1; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  unset ZF_LIB_PATH; [ "$lib" != "-" ] && export ZF_LIB_PATH=$lib
  rm -rf /tmp/tr_okt
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_okt -o t -- python3 $ROOT/tools/op_bench.py --size $SZ --iters 100 > /dev/null 2>&1
  t=$(find /tmp/tr_okt -name "*kernel_trace.csv" | head -1)
  python3 - "$t" "$lib" $SZ <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "zf_op_" in n or "zf_trial_kernel" in n:
        d[n.split("<")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, v in d.items():
    v.sort()
    busy = [x for x in v if x > 0.5 * v[-len(v) // 4]]
    out[k] = (len(busy), round(sorted(busy)[len(busy) // 2] / 1e3, 1))
print(sys.argv[3], sys.argv[2].split("/")[-1], out, "sum", round(sum(x[1] for x in out.values()), 1))
PY
done
