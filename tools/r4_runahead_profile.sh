#!/bin/bash
# kernel traces of the clean regime at n = 1e7 and 1e6 with run-ahead passes and with one launch per pass:
#   tools/r4_runahead_profile.sh OUTDIR
set -e
OUT="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; mkdir -p "$OUT"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp; export TMPDIR=/tmp
for n in 10000000 1000000; do
  for ra in 1 0; do
    export ZF_RUNAHEAD=$ra
    D=$OUT/trace_n${n}_ra${ra}
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python3 $ROOT/bench.py --n $n --steps 64 --warmup 16 --min-seconds 0.2 --no-cpu-baseline --no-regimes > $OUT/bench_n${n}_ra${ra}_under_rocprof.json 2>$OUT/err_n${n}_ra${ra}.txt
    python3 $ROOT/bench.py --n $n --steps 64 --warmup 16 --no-cpu-baseline --no-regimes > $OUT/bench_n${n}_ra${ra}.json 2>>$OUT/err_n${n}_ra${ra}.txt
    if [ $ra = 1 ]; then python3 $ROOT/tools/runahead_timeline.py $D > $OUT/timeline_n${n}.json; else python3 $ROOT/tools/timeline.py $D 1 > $OUT/timeline_n${n}_per_pass.json; fi
    cp $(find $D -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_n${n}_ra${ra}.csv
    rm -rf $D
  done
done
unset ZF_RUNAHEAD
ls $OUT
