#!/usr/bin/env python3
"""Cost of a tail pass by its length: blocks of K = 2 L iterations from iteration 5 are two passes of L trials
(zf_fresh_len) - L = 9 .. 15 through the branch-free mid chain of that length, and through the general body
(ZF_MID_CHAINS=0), alternating in one process.  Also K = 16 (one full chain) and K = 8.   tools/r4_tails.py [n]"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LAM, LR, make_inputs  # noqa: E402
from zfista_amd import _lib  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
W = 5
d, c = make_inputs(n, 1, "cuda")
prob = DiagQuadL1(d, c, LAM)
x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
o = dict(lr=LR, tol=0.0, tol_internal=1e-12, max_iter=W, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
         nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=16)


def block(K):
    run = NativeRun(prob, x0, dict(o, max_iter=W), timing=True)
    S = run.sub_iters
    while run.status == _lib.ZF_RUNNING:
        run.advance(1)
    run.solver.pass_stats()
    run.set_max_iter(W + K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while run.status == _lib.ZF_RUNNING:
        run.enqueue_only((W + K - run.nit_seen + S - 1) // S)
        run.collect()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    (fm, fn), (pm, pn), (pf, pl) = run.solver.pass_stats_ex()
    run.solver.close()
    return dt, (fm, fn, pm, pn, pf)


rows = []
for L in list(range(9, 16)) + [16, 8, 4]:
    K = 2 * L if L < 16 and L > 8 else L
    acc = {"mid": [], "general": []}
    ker = {}
    for rep in range(24):
        for mode in ("mid", "general"):
            if mode == "general":
                os.environ["ZF_MID_CHAINS"] = "0"
            else:
                os.environ.pop("ZF_MID_CHAINS", None)
            dt, st = block(K)
            acc[mode].append(dt)
            ker.setdefault(mode, []).append(st[2] if st[3] else st[0])
    os.environ.pop("ZF_MID_CHAINS", None)
    row = dict(n=n, trials_per_pass=L, K=K)
    for mode in acc:
        row[mode] = dict(block_ms=statistics.median(acc[mode][4:]) * 1e3, kernel_ms_per_pass=statistics.median(ker[mode][4:]),
                         hbm_frac=48.0 * n / (statistics.median(ker[mode][4:]) * 1e-3) / 8e12)
    print(json.dumps(row), flush=True)
