#!/usr/bin/env python3
"""Cost of return_all on the device-resident path (n = 1e7 by default): the same K FISTA iterations
without recording, with the iterates recorded into the HBM ring by the trial kernel (chains of 8),
and - for scale - the round-1 way (one iteration per pass + a blocking download per iteration).
        tools/return_all_rate.py [n] [K]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LAM, LR, make_inputs  # noqa: E402
from zfista_amd import _lib  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**7
K = int(sys.argv[2]) if len(sys.argv) > 2 else 80
d, c = make_inputs(n, 1, "cuda")
prob = DiagQuadL1(d, c, LAM)
x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
base = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
            nesterov_ratio=(0, 0.25), deprecated=False)


def run(**kw):
    best = None
    for _ in range(3):
        r = NativeRun(prob, x0, dict(base, **kw))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while r.status == _lib.ZF_RUNNING:
            r.advance(64)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        hist = r.history() if kw.get("return_all") else None
        last = None if hist is None else hist[len(hist) - 1]
        xk = r.solver.get_x()
        r.solver.close()
    return best, xk, last


t_plain, x_plain, _ = run()
t_rec, x_rec, last = run(return_all=True)
assert np.array_equal(x_plain, x_rec) and np.array_equal(last, x_rec)
t_rec1, _, _ = run(return_all=True, sub_iters=1)
# the round-1 way: one pass per iteration and a blocking download of every iterate
r = NativeRun(prob, x0, dict(base, sub_iters=1))
torch.cuda.synchronize()
t0 = time.perf_counter()
while r.status == _lib.ZF_RUNNING:
    r.advance(1)
    r.solver.get_x()
torch.cuda.synchronize()
t_old = time.perf_counter() - t0
r.solver.close()
print(json.dumps(dict(n=n, iterations=K, it_per_s_not_recording=K / t_plain, it_per_s_recording_chain8=K / t_rec,
                      it_per_s_recording_chain1=K / t_rec1, it_per_s_round1_download_per_iteration=K / t_old,
                      recording_slowdown=t_rec / t_plain, ring_GB=8e-9 * n * (K + 1))))
