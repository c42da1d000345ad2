#!/usr/bin/env python3
"""Pass accounting of a LONG solve of the headline problem: how many passes K iterations take
once chains start breaking at the resolution limit of the acceptance test, and
what the breaks cost.        tools/long_run.py [n] [K] [S]
Prints one JSON line: iterations, rejections, passes (full chains / others), ideal passes
ceil(K / S), kernel time by pass kind, it/s."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LAM, LR, make_inputs  # noqa: E402
from zfista_amd import _lib  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

ACCEPT = "reference"
if "--acceptance" in sys.argv:   # long_run.py [n] [K] --acceptance resolved : the acceptance test resolved below ulp(F)
    at = sys.argv.index("--acceptance")
    ACCEPT = sys.argv[at + 1]
    del sys.argv[at:at + 2]
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
S = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d, c = make_inputs(n, 1, "cuda")
o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
         nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=S, acceptance=ACCEPT)
x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
for rep in range(2):   # second repetition on a warm device is the one reported
    run = NativeRun(DiagQuadL1(d, c, LAM), x0, o, timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while run.status == _lib.ZF_RUNNING:
        run.advance(64)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    records = run.solver.pass_records()
    (fm, fn), (pm, pn) = run.solver.pass_stats()
    ctl = run.solver.ctl
    S_used = run.sub_iters
    out = dict(n=n, acceptance=ACCEPT, iterations=int(ctl.nit), chain=S_used, rejections=int(ctl.total_trials - ctl.nit), lr_final=ctl.lr,
               passes=fn + pn, full_chain_passes=fn, other_passes=pn, ideal_passes=-(-K // S_used),
               extra_passes_frac=(fn + pn) / (-(-K // S_used)) - 1.0,
               full_chain_ms=fm, other_ms=pm, kernel_ms_total=fm * fn + pm * pn,
               kernel_time_over_ideal=(fm * fn + pm * pn) / max(fm * (-(-K // S_used)), 1e-30) - 1.0 if fn else None,
               seconds=dt, it_per_s=int(ctl.nit) / dt)
    run.solver.close()
# kernel time by pass shape (lagging iterations replayed, fresh trials): count, mean ms
by_shape = {}
for lag, nf, cnt, ms in records:
    by_shape.setdefault((lag, nf), []).append(ms)
out["by_shape"] = [dict(lag=k[0], fresh=k[1], passes=len(v), mean_ms=sum(v) / len(v)) for k, v in sorted(by_shape.items())]
print(json.dumps(out))
