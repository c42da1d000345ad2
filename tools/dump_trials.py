#!/usr/bin/env python3
"""Trials per accepted iteration of the headline solve (the acceptance sequence is the same for every chain length):
input of tools/sim_policy.py.   tools/dump_trials.py n K"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LAM, LR, make_inputs  # noqa: E402
from zfista_amd import _lib  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

n, K = int(float(sys.argv[1])), int(sys.argv[2])
d, c = make_inputs(n, 1, "cuda")
o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
         nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=16)
run = NativeRun(DiagQuadL1(d, c, LAM), torch.zeros(n, dtype=torch.float64, device="cuda"), o)
rows = []
while run.status == _lib.ZF_RUNNING:
    rows.append(run.advance(8))
rows = np.concatenate(rows)
print(json.dumps(dict(n=n, K=K, trials=[int(t) for t in rows[:, _lib.TR_TRIALS]])))
