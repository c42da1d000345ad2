#!/usr/bin/env python3
"""The momentum sweep of the reference's ``examples/cameraman.ipynb`` (cells 10-12) on one GPU: the deblurring LASSO
(9 x 9 Gaussian blur, one Haar level, 256 x 256 synthetic image - scikit-image's ``camera()`` is not in this image) solved
for the notebook's 15 momentum settings (a, b) with its call - ``lr = 1 / L, decay_rate = 1, nesterov = True,
return_all = True``, default ``tol`` - one after the other, and all at once on 15 HIP streams
(``zfista_amd.replicas.solve_on_streams``).  The reference's recorded run: ``Parallel(n_jobs=-1)``, 8 loky workers,
"15 out of 15 | elapsed: 5.8min" (cell 11), 318-517 iterations per setting (``examples/data/cameraman_ab.csv``).

    python examples/deblur_sweep.py [--size 256] [--streams 15] [--check]

Prints one JSON line; --check also runs the CPU oracle on the first and the last setting (iterates to 1e-10)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings
from fractions import Fraction as Fr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

RATIOS = [(0, 0), (0, Fr(1, 8)), (0, Fr(1, 4)), (Fr(1, 6), Fr(1, 144)), (Fr(1, 6), Fr(37, 288)), (Fr(1, 6), Fr(1, 4)),
          (Fr(1, 4), Fr(1, 64)), (Fr(1, 4), Fr(17, 128)), (Fr(1, 4), Fr(1, 4)), (Fr(1, 2), Fr(1, 16)), (Fr(1, 2), Fr(5, 32)),
          (Fr(1, 2), Fr(1, 4)), (Fr(3, 4), Fr(9, 64)), (Fr(3, 4), Fr(25, 128)), (Fr(3, 4), Fr(1, 4))]   # cameraman.ipynb cell 10


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--streams", type=int, default=15)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    import torch

    from oracle.operator_ref import L1_RATIO, BlurHaarL1Ref, make_deblur
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import BlurHaarL1
    from zfista_amd.replicas import solve_on_streams

    kernel, observed, x0, L = make_deblur(a.size)
    prob = BlurHaarL1(kernel, observed, L1_RATIO)
    kws = [dict(lr=1 / L, decay_rate=1, nesterov=True, nesterov_ratio=tuple(map(float, r)), return_all=True) for r in RATIOS]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        minimize_proximal_gradient(*prob.callbacks(), x0, max_iter=5, **kws[0])   # warm-up (kernel caches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seq = [minimize_proximal_gradient(*prob.callbacks(), x0, **kw) for kw in kws]
        torch.cuda.synchronize()
        t_seq = time.perf_counter() - t0
        t0 = time.perf_counter()
        par = solve_on_streams([(prob, x0, kw) for kw in kws], streams=a.streams)
        torch.cuda.synchronize()
        t_par = time.perf_counter() - t0
    same = all(p.nit == s.nit and np.array_equal(p.x, s.x) for p, s in zip(par, seq))
    out = {"workload": f"momentum sweep of examples/cameraman.ipynb: 15 settings, {a.size} x {a.size}, default tol, return_all",
           "iterations_per_setting": [int(r.nit) for r in seq], "total_iterations": int(sum(r.nit for r in seq)),
           "wall_s_one_after_the_other": t_seq, "wall_s_on_streams": t_par, "streams": a.streams,
           "it_per_s_one_after_the_other": sum(r.nit for r in seq) / t_seq, "it_per_s_on_streams": sum(r.nit for r in par) / t_par,
           "results_identical": bool(same), "reference_recorded_wall_s": 5.8 * 60,
           "reference_iterations_per_setting": [517, 517, 517, 480, 480, 480, 417, 417, 416, 319, 319, 318, 397, 391, 387]}
    if a.check:
        from oracle import cpu_ref

        ref = BlurHaarL1Ref(kernel, observed)
        errs = []
        for k in (0, len(kws) - 1):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **kws[k])
            errs.append(float(np.linalg.norm(seq[k].x - exp.x) / np.linalg.norm(exp.x)))
            assert exp.nit == seq[k].nit, (exp.nit, seq[k].nit)
        out["rel_err_vs_cpu_oracle"] = errs
    print(json.dumps(out))


if __name__ == "__main__":
    main()
