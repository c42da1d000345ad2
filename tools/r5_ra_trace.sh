#!/bin/bash
# kernel timeline of run-ahead blocks at n = 1e6 (K = 64: four full chains; K = 20: two mid chains), and where a run that
# starts with a mid chain begins to pay (K = 20 blocks by size)
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_ra_mid
for K in 64 20; do
  rm -rf /tmp/tr_$K
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr_$K -o t -- python3 tools/block_probe.py --n 1000000 --steps $K --reps 30 > /dev/null 2>&1
  f=$(find /tmp/tr_$K -name "*kernel_trace.csv" | head -1)
  m=$(find /tmp/tr_$K -name "*memory_copy_trace.csv" | head -1)
  python3 - "$f" "$m" $K <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?") + " " + r["Kernel_Name"][:60]) for r in csv.DictReader(open(sys.argv[1]))]
try:
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")) for r in csv.DictReader(open(sys.argv[2]))]
except Exception as e:
    print("no copy trace", e)
rows.sort()
tail = rows[-40:]
t0 = tail[0][0]
out = open(f"gpurun_out/r5_ra_mid/timeline_n1e6_K{sys.argv[3]}.txt", "w")
for a, b, name in tail:
    out.write(f"{(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f} us  {(b - a) / 1e3:7.1f}  {name}\n")
out.close()
print(open(f"gpurun_out/r5_ra_mid/timeline_n1e6_K{sys.argv[3]}.txt").read())
PY
done
out=gpurun_out/r5_ra_mid/k20_by_size.jsonl; : > $out
for n in 1000000 2000000 4000000 6000000 10000000 20000000; do
  for ra in 0 1; do
    echo -n "ra=$ra " >> $out
    ZF_RUNAHEAD=$ra timeout -k 10 300 python tools/block_probe.py --n $n --steps 20 --reps 40 2>/dev/null >> $out || exit 1
  done
done
cat $out
