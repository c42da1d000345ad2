#!/usr/bin/env python3
"""Soak: many random sizes (single- and multi-workgroup finalize, ragged tails, autotuned tiles)
and option sets; the chained solves (S = 8 and S = 16) must reproduce the one-trial-per-pass solve bit for bit,
and two runs of the same solve must be identical (no timing-dependent reduction anywhere)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zfista_amd import _lib  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(42)


def solve(prob, x0, o, sub):
    run = NativeRun(prob, x0, dict(o, sub_iters=sub))
    rows = []
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(int(rng.integers(1, 9))))
    out = (np.concatenate(rows) if rows else np.zeros((0, 8)), run.solver.get_x(), int(run.solver.ctl.nit),
           int(run.solver.ctl.status), run.solver.ctl.lr)
    run.solver.close()
    return out


bad = 0
for k in range(cases):
    n = int(10 ** rng.uniform(2, 6.9))
    gen = torch.Generator(device="cuda").manual_seed(k)
    d = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 1.5 + 0.5
    c = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
    o = dict(lr=float(10 ** rng.uniform(-1, 1.5)), tol=float(rng.choice([0.0, 1e-6])), tol_internal=1e-12,
             max_iter=int(rng.integers(5, 70)), max_backtrack_iter=100, decay_rate=float(rng.choice([0.5, 0.8])),
             nesterov=bool(rng.integers(0, 2)), nesterov_ratio=(0, 0.25), deprecated=False)
    prob = DiagQuadL1(d, c, 0.1)
    x0 = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
    a, b, a2, c16 = solve(prob, x0, o, 1), solve(prob, x0, o, 8), solve(prob, x0, o, 8), solve(prob, x0, o, 16)
    same = all(np.array_equal(u, v) if isinstance(u, np.ndarray) else u == v for u, v in zip(a, b)) and \
        all(np.array_equal(u, v) if isinstance(u, np.ndarray) else u == v for u, v in zip(a, c16))
    rep = all(np.array_equal(u, v) if isinstance(u, np.ndarray) else u == v for u, v in zip(b, a2))
    if not (same and rep):
        bad += 1
        print(f"case {k}: n={n} {o} chained==single: {same}, repeatable: {rep}", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
