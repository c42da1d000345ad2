#!/bin/bash
# kernel timeline of passes ahead (unsharded scheme, n = 1e7, K = 64): where do the gaps between the trial kernels come from
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_ahead
for mode in ahead perpass libcomm; do
  case $mode in
    ahead) e="ZF_RUNAHEAD=0 ZF_AHEAD_UNSHARDED=1"; extra="";;
    perpass) e="ZF_RUNAHEAD=0 ZF_AHEAD_UNSHARDED=0"; extra="";;
    libcomm) e="ZF_X=1"; extra="--libcomm";;
  esac
  rm -rf /tmp/tr_$mode
  env $e rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$mode -o t -- python3 bench.py --n 10000000 --steps 64 --warmup 16 --min-seconds 0.05 --no-cpu-baseline --no-regimes --no-kernel-events $extra > /dev/null 2>&1
  f=$(find /tmp/tr_$mode -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $mode <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the middle of the run: 60 consecutive kernels
mid = rows[len(rows) // 2: len(rows) // 2 + 60]
t0 = int(mid[0]["Start_Timestamp"])
out = open(f"gpurun_out/r5_ahead/timeline_{sys.argv[2]}.txt", "w")
for r in mid:
    name = r["Kernel_Name"][:70]
    out.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {(int(r['End_Timestamp']) - t0) / 1e3:10.1f} us  q{r.get('Queue_Id', '?')}  {name}\n")
out.close()
print(open(f"gpurun_out/r5_ahead/timeline_{sys.argv[2]}.txt").read()[:3500])
PY
done
