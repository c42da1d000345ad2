#!/bin/bash
# the whole GPU suite with the round's build, smoke(), the operator rates again, the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_suite
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5_suite/gpu_suite.log 2>&1; rc=$?
tail -4 gpurun_out/r5_suite/gpu_suite.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for sz in 256 1024 4096; do python tools/op_bench.py --size $sz --iters 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n'], round(d['it_per_s'],1), 'it/s', round(d['ms_per_iteration'],4), 'ms')"; done
(time python bench.py > gpurun_out/r5_suite/bench_default.json 2> gpurun_out/r5_suite/bench_default.err) 2>&1 | tail -3
python -c "import json; d=json.load(open('gpurun_out/r5_suite/bench_default.json')); print(d['value'], d['roofline']['bound'], d['roofline']['frac'], d['cpu_baseline']['value'])"
