#!/usr/bin/env python3
"""Soak of the run-ahead passes: random sizes (one workgroup ... one round of 489), options (momentum, step sizes that
are rejected first, tolerances that end a solve inside a chain), chunkings of the host loop and - to force the void /
poison paths - random tiny spin limits; every solve must equal the one-launch-per-pass solve (ZF_RUNAHEAD=0) bit for bit.
    tools/soak_runahead.py [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zfista_amd import _lib  # noqa: E402
from zfista_amd.problems import DiagQuadL1  # noqa: E402
from zfista_amd.proximal_gradient import NativeRun  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(7)


def solve(prob, x0, o, chunks):
    run = NativeRun(prob, x0, dict(o, sub_iters=16))
    rows, k = [], 0
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(chunks[k % len(chunks)]))
        k += 1
    out = (np.concatenate(rows) if rows else np.zeros((0, 8)), run.solver.get_x(), run.solver.get_x_prev(), int(run.solver.ctl.nit),
           int(run.solver.ctl.status), run.solver.ctl.lr, int(run.solver.ctl.total_trials))
    ra = run.solver.runahead_counts()
    run.solver.close()
    return out, ra


bad = ahead_total = 0
for case in range(cases):
    n = int(rng.choice([rng.integers(2000, 40000), rng.integers(40000, 400000), rng.integers(400000, 3000000)]))
    d = rng.uniform(0.5, 2.0, n)
    c = rng.standard_normal(n)
    lam = float(rng.choice([0.0, 0.05, 0.3]))
    prob = DiagQuadL1(d, c, lam)
    x0 = np.zeros(n) if rng.random() < 0.5 else rng.standard_normal(n)
    o = dict(lr=float(rng.choice([0.45, 0.3, 1.0, 7.0])), tol=float(rng.choice([0.0, 0.0, 1e-6, 1e-9])), tol_internal=1e-12,
             max_iter=int(rng.integers(17, 400)), max_backtrack_iter=100, decay_rate=float(rng.choice([0.5, 0.7])),
             nesterov=bool(rng.random() < 0.8), nesterov_ratio=(0, 0.25) if rng.random() < 0.7 else (0.5, 1 / 16), deprecated=False)
    chunks = [int(v) for v in rng.integers(1, 12, size=4)]
    os.environ["ZF_RUNAHEAD"] = "0"
    ref, _ = solve(prob, x0, o, chunks)
    del os.environ["ZF_RUNAHEAD"]
    lim = None if rng.random() < 0.5 else str(int(rng.integers(0, 40)))
    if lim is None:
        os.environ.pop("ZF_RUNAHEAD_SPIN_LIMIT", None)
    else:
        os.environ["ZF_RUNAHEAD_SPIN_LIMIT"] = lim
    got, ra = solve(prob, x0, o, chunks)
    os.environ.pop("ZF_RUNAHEAD_SPIN_LIMIT", None)
    same = all(np.array_equal(u, v) if isinstance(u, np.ndarray) else u == v for u, v in zip(ref, got))
    ahead_total += ra[1]
    if not same:
        bad += 1
        print("MISMATCH", case, n, o, chunks, "spin limit", lim, "nit", ref[3], got[3], "trials", ref[6], got[6], flush=True)
print(f"{cases} cases, {bad} mismatches, {ahead_total} passes launched behind a pass in flight")
sys.exit(1 if bad else 0)
