#!/usr/bin/env python3
"""Print the control block after every pass of the headline solve (n = 1e8 by default): where
chains break (rejections at the noise floor of the acceptance test) and what that costs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LAM, LR, make_inputs  # noqa: E402
from zfista_amd import _lib  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 110
d, c = make_inputs(n, 1, "cuda")
o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
         nesterov_ratio=(0, 0.25), deprecated=False)
run = NativeRun(DiagQuadL1(d, c, LAM), torch.zeros(n, dtype=torch.float64, device="cuda"), o)
p = 0
while run.status == _lib.ZF_RUNNING:
    rows = run.advance(1)
    p += 1
    ctl = run.solver.ctl
    extra = f"   <- chain broke: {ctl.lag} accepted iterations wait for the next pass to materialise" if ctl.lag else ""
    print(f"pass {p:3d}  nit {ctl.nit:4d}  (+{len(rows)})  lr {ctl.lr:.6g}  trials {ctl.total_trials:4d}  lag {ctl.lag}  "
          f"err {ctl.err:.3e}  F {ctl.F_old:.17g}{extra}")
