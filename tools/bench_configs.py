#!/usr/bin/env python3
"""Secondary measurements for the non-headline BASELINE configs (one JSON line each).

  cfg1  LASSO 512 x 1024 (launch-bound plumbing case)            --cfg 1
  cfg2  P-diag n = 1e7                                            --cfg 2
  cfg3  dense LASSO 16384 x 65536 fp64 (8 GiB A, HBM-bound GEMVs) --cfg 3
  cfg4  FDS m = 3, n = 1e6 (host dual solver + device dual evals) --cfg 4

bench.py stays the driver's contract for the headline metric; this tool feeds
DESIGN.md / profiles with the other rows of SURVEY.md 8(d).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _opts(**kw):
    o = dict(lr=1, tol=0.0, tol_internal=1e-12, max_iter=100, max_backtrack_iter=100, decay_rate=0.5,
             nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False)
    o.update(kw)
    return o


def run_native(prob, x0, opts, K, W, _warm=True):
    """Exactly W untimed then K timed accepted iterations (max_iter is raised from W to W + K and
    the device stops on it), as bench.py does.  The same solve runs once untimed first: every shape-specific kernel is
    loaded on its first launch (~0.5 ms each - round 4 has one code object per kernel group), which a single cold run
    of a few milliseconds would count as solve time."""
    import torch

    if _warm:
        run_native(prob, x0, opts, K, W, _warm=False)

    from zfista_amd import _lib
    from zfista_amd.proximal_gradient import NativeRun

    run = NativeRun(prob, x0, dict(opts, max_iter=max(W, 1)), timing=True)
    S = run.sub_iters
    while W > 0 and run.status == _lib.ZF_RUNNING:
        run.advance((W - run.nit_seen + S - 1) // S)
    run.solver.trial_kernel_ms()
    nit0, trials0 = run.nit_seen, int(run.solver.ctl.total_trials)
    run.set_max_iter(nit0 + K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while run.status == _lib.ZF_RUNNING:
        # (as bench.py: two spare passes per round once the block has seen rejections)
        spare = 2 if int(run.solver.ctl.total_trials) - run.nit_seen > trials0 - nit0 else 0
        run.enqueue_only((K - (run.nit_seen - nit0) + S - 1) // S + spare)
        run.collect()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc = run.nit_seen - nit0
    ms, cnt = run.solver.trial_kernel_ms()
    ctl = run.solver.ctl
    out = dict(iterations=acc, seconds=dt, it_per_s=acc / dt, trial_kernel_ms=ms, passes=cnt, chain=S,
               tiles_per_wg=getattr(run.solver, "tiles_per_wg", None),
               lr_final=ctl.lr, trials=int(ctl.total_trials) - trials0, status=int(ctl.status))
    run.solver.close()
    return out


def lasso(m, n, seed, K, W):
    import torch

    from zfista_amd.problems import LeastSquaresL1

    gen = torch.Generator(device="cuda").manual_seed(seed)
    A = torch.randn(m, n, dtype=torch.float64, device="cuda", generator=gen)
    x_true = torch.zeros(n, dtype=torch.float64, device="cuda")
    x_true[:20] = torch.randn(20, dtype=torch.float64, device="cuda", generator=gen)
    b = A @ x_true + 0.01 * torch.randn(m, dtype=torch.float64, device="cuda", generator=gen)
    lam = 0.1 * float(torch.max(torch.abs(A.T @ b)))
    # lr = 0.9 / |A|_2^2 by 20 power iterations (input preparation, not the timed path)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
    for _ in range(20):
        v = A.T @ (A @ v)
        v /= torch.linalg.norm(v)
    L = float(torch.linalg.norm(A @ v)) ** 2
    prob = LeastSquaresL1(A, b, lam, scale=0.5)
    # f = 1/2 |Ax-b|^2 has Lipschitz constant |A|_2^2
    r = run_native(prob, torch.zeros(n, dtype=torch.float64, device="cuda"), _opts(lr=0.9 / L, max_iter=K + W), K, W)
    bytes_per_iter = 2.0 * m * n * 8
    r.update(workload=f"dense LASSO {m}x{n} fp64, lr=0.9/L", algorithmic_bytes_per_iteration=bytes_per_iter,
             achieved_GBps_whole_iteration=bytes_per_iter * r["it_per_s"] / 1e9,
             frac_of_8TBps=bytes_per_iter * r["it_per_s"] / 8e12)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, required=True)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dual-solver", default=os.environ.get("ZF_DUAL_SOLVER", "scipy"))
    a = ap.parse_args()
    import torch

    if a.cfg == 1:
        from oracle import problems_ref as P
        from zfista_amd.problems import LeastSquaresL1

        A, b, lam = P.make_plasso(512, 1024, seed=0)
        prob = LeastSquaresL1(A, b, lam)
        r = run_native(prob, np.zeros(1024), _opts(lr=1, max_iter=a.steps + a.warmup), a.steps, a.warmup)
        r["workload"] = "cfg1 LASSO 512x1024, lr=1 (backtracks to 2^-11 in iteration 1)"
    elif a.cfg == 2:
        from zfista_amd.problems import DiagQuadL1

        n = 10**7
        gen = torch.Generator(device="cuda").manual_seed(1)
        d = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 1.5 + 0.5
        c = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
        r = run_native(DiagQuadL1(d, c, 0.1), torch.zeros(n, dtype=torch.float64, device="cuda"),
                       _opts(lr=0.45, max_iter=a.steps + a.warmup), a.steps, a.warmup)
        r.update(workload="cfg2 P-diag n=1e7",
                 algorithmic_GBps_trial_kernel=40.0 * n * r["iterations"] / r["passes"] / r["trial_kernel_ms"] / 1e6,
                 hbm_GBps_trial_kernel=(48.0 if r["chain"] > 1 else 40.0) * n / r["trial_kernel_ms"] / 1e6)
    elif a.cfg == 3:
        r = lasso(16384, 65536, 3, a.steps, a.warmup)
    elif a.cfg == 4:
        from zfista_amd import minimize_proximal_gradient
        from zfista_amd.problems import FDS

        n = 10**6
        prob = FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0, 1, 2])
        x0 = np.random.default_rng(1).uniform(-2, 2, n)
        native = a.dual_solver in ("native", "device")
        K = a.steps if native else min(a.steps, 10)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            minimize_proximal_gradient(*prob.callbacks(), x0, lr=1e-7, nesterov=True, tol=0.0, max_iter=2,
                                       dual_solver=a.dual_solver)  # warm-up
            prob._engine().n_dual_evals = 0
            t0 = time.perf_counter()
            res = minimize_proximal_gradient(*prob.callbacks(), x0, lr=1e-7, nesterov=True, tol=0.0, max_iter=K,
                                             dual_solver=a.dual_solver)
            dt = time.perf_counter() - t0
        ev = prob._engine().n_dual_evals
        r = dict(workload="cfg4 FDS m=3 n=1e6 + l1, lr=1e-7, dual solver: " + (a.dual_solver + " (library simplex Newton)" if native else
                                                                                   "SciPy trust-constr (reference)"),
                 iterations=int(res.nit), seconds=dt,
                 it_per_s=res.nit / dt, dual_evals=ev, dual_evals_per_s=ev / dt,
                 bytes_per_dual_eval=8 * 4 * n)
    else:
        raise SystemExit("cfg must be 1..4")
    r["cfg"] = a.cfg
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
