#!/usr/bin/env python3
"""Bit-identity of the solve across chain lengths 1 / 8 / 16 at sizes where the tiles per workgroup follow whole
rounds (n >= 3.4e7: T = 11 .. 24), odd n, with backtracking line searches."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zfista_amd import _lib
from zfista_amd.problems import DiagQuadL1
from zfista_amd.proximal_gradient import NativeRun
def solve(prob, x0, o, sub):
    run = NativeRun(prob, x0, dict(o, sub_iters=sub)); rows = []
    while run.status == _lib.ZF_RUNNING: rows.append(run.advance(3))
    out = (np.concatenate(rows), run.solver.get_x(), int(run.solver.ctl.nit), int(run.solver.ctl.status), run.solver.ctl.lr, getattr(run.solver, "tiles_per_wg", None))
    run.solver.close(); return out
for n, lr, K in [(50_000_003, 4.0, 37), (34_000_000, 0.45, 45), (120_000_000, 3.0, 29)]:
    gen = torch.Generator(device="cuda").manual_seed(n % 1000)
    d = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 1.5 + 0.5
    c = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
    o = dict(lr=lr, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5, nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False)
    prob = DiagQuadL1(d, c, 0.1); x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    a, b, c8 = solve(prob, x0, o, 1), solve(prob, x0, o, 16), solve(prob, x0, o, 8)
    same = all(np.array_equal(u, v) if isinstance(u, np.ndarray) else u == v for u, v in zip(a[:5], b[:5])) and all(np.array_equal(u, v) if isinstance(u, np.ndarray) else u == v for u, v in zip(a[:5], c8[:5]))
    print(n, "nit", a[2], "lr_final", a[4], "S1==S16==S8:", same, flush=True)
    del d, c, x0, prob; torch.cuda.empty_cache()
