#!/bin/bash
# Same-box A/B of the headline line (bench.py --steps 20 --warmup 5, n = 1e8): the round-3 tree (exported at 67c0133 under
# _ab/r3tree, built there) against this tree, alternating, three repetitions each.  Output: one JSON line per run with
# its tree in front -> gpurun_out/r5_ab/headline.txt (copied to profiles/r05_headline_r3_vs_head_same_box.txt)
out=$PWD/gpurun_out/r5_ab; mkdir -p $out
: > $out/headline.txt
for rep in 1 2 3; do
  for tree in r3 head; do
    if [ $tree = r3 ]; then dir=$PWD/_ab/r3tree; else dir=$PWD; fi
    line=$(cd $dir && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes 2>/dev/null | tail -1)
    echo "$tree $line" >> $out/headline.txt
    echo "$tree rep $rep done"
  done
done
python - <<'EOF'
import json
rows = [l.split(" ", 1) for l in open("gpurun_out/r5_ab/headline.txt") if l.strip()]
for tree, js in rows:
    try:
        d = json.loads(js)
    except Exception as e:
        print(tree, "unparsed:", js[:200]); continue
    r = d.get("roofline", {})
    print(f"{tree:5s} value {d['value']:9.1f} it/s  ms_per_step {d['ms_per_step']:.5f}  kernel_avg_ms {r.get('kernel_avg_ms')}  frac {r.get('frac')}")
EOF
