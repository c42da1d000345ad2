#!/usr/bin/env python3
"""Pareto-front experiment harness: the experimental design of the reference's
``benchmarks/benchmark.py`` (:303-376 the three solver variants per random start; :413-470 the
problem list and the start ranges) on this engine, with ``zfista_amd.metrics`` for the tables.
No plots; results go to a JSON file.  The reference's three ``Parallel(n_jobs=-1)`` sweeps over the
start points (:325,:341,:360) are replicas: ``--workers W`` spreads them over W worker processes,
one per GPU (``zfista_amd.replicas.solve_replicas``; W > GPUs: the workers share devices).

    python tools/harness.py --samples 20 --problems JOS1,SD,FDS --workers 8 --out harness.json

Variants per start x0 ~ U[low, high]^n (as the reference): proximal gradient, accelerated
(``nesterov=True``), accelerated with the ``deprecated`` acceptance test; all with
``tol_internal=1e-11`` and the default ``tol=1e-5``.
"""
from __future__ import annotations

import argparse
import inspect
import json
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

RANGES = {   # benchmarks/benchmark.py:460-470
    "JOS1": (-2, 4), "FDS": (-2, 2), "SD": ([1, np.sqrt(2), np.sqrt(2), 1], [3, 3, 3, 3]), "ZDT1": (0, 0.01),
    "TOI4": (-2, 5), "TRIDIA": (-1, 1), "LinearFunctionRank1": (-1, 1),
}
SIZES = {"JOS1": [5, 10, 20, 50, 100, 200, 500, 1000], "ZDT1": [50, 100], "FDS": [5, 10, 20, 50, 100],
         "LinearFunctionRank1": [30]}   # :425-430


def build_problems(names, max_n):
    from zfista_amd import problems as Z

    out = []
    for name in names:
        cls = getattr(Z, name)
        params = inspect.signature(cls.__init__).parameters
        has_l1 = "l1_ratios" in params and "l1_shifts" in params
        for n in SIZES.get(name, [None]):
            if n is not None and n > max_n:
                continue
            kw = {} if n is None else dict(n_features=n)
            prob = cls(**kw)
            out.append(prob)
            if has_l1:   # the regularised twin: l1_ratios = (1..m) / n, l1_shifts = 0..m-1   (:437-452)
                m, nf = prob.n_objectives, prob.n_features
                out.append(cls(l1_ratios=(np.arange(m) + 1) / nf, l1_shifts=np.arange(m), **kw))
    return out


def run_variants(problem, starts, max_iter, tol_internal, workers=1, dual_solver="scipy"):
    variants = {"Normal": {}, "Accelerated": dict(nesterov=True),
                "Accelerated (deprecated test)": dict(nesterov=True, deprecated=True)}
    results = {}
    for label, kw in variants.items():
        common = dict(return_all=True, max_iter=max_iter, tol_internal=tol_internal, dual_solver=dual_solver, **kw)
        if workers > 1:
            from zfista_amd.replicas import ProblemRecipe, solve_replicas

            results[label] = solve_replicas(ProblemRecipe(problem), list(starts), workers=workers, **common)
            continue
        rs = []
        for x0 in starts:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                rs.append(problem.minimize_proximal_gradient(x0, **common))
        results[label] = rs
    return results


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=20, help="random starts per problem (the reference: 100-1000)")
    ap.add_argument("--problems", default="JOS1,SD,TOI4,TRIDIA,LinearFunctionRank1,ZDT1,FDS")
    ap.add_argument("--max-n", type=int, default=100, help="skip problem sizes above this")
    ap.add_argument("--max-iter", type=int, default=100000000)
    ap.add_argument("--tol-internal", type=float, default=1e-11)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--workers", type=int, default=1, help="replica worker processes (one per GPU)")
    ap.add_argument("--dual-solver", default="scipy", help="scipy (the reference's calls) | native | device")
    ap.add_argument("--out", default="harness.json")
    a = ap.parse_args(argv)
    from zfista_amd.metrics import calculate_metrics

    rng = np.random.default_rng(a.seed)
    report = {}
    for prob in build_problems(a.problems.split(","), a.max_n):
        low, high = RANGES[type(prob).__name__]
        starts = rng.uniform(low, high, size=(a.samples, prob.n_features))
        t0 = time.time()
        res = run_variants(prob, starts, a.max_iter, a.tol_internal, a.workers, a.dual_solver)
        metrics, ratios = calculate_metrics(*res.items())
        ok = {k: [r for r in v if r.success] for k, v in res.items()}
        report[prob.name] = {
            "metrics": {k: {n: (None if v is None or np.isnan(v) else float(v)) for n, v in d.items()}
                        for k, d in metrics.items()},
            "ratios": {k: {n: float(v) for n, v in d.items()} for k, d in ratios.items()},
            "iterations_per_second": {k: (float(sum(r.nit for r in v) / max(sum(r.time for r in v), 1e-300)) if v else None)
                                      for k, v in ok.items()},
            "wall_seconds": time.time() - t0,
        }
        line = ", ".join(f"{k}: {report[prob.name]['metrics']['Avg iterations'][k]:.1f} it" for k in res)
        print(f"{prob.name:48s} {time.time() - t0:6.1f} s   {line}", flush=True)
    with open(a.out, "w") as fh:
        json.dump(report, fh, indent=1)
    print("wrote", a.out)
    return report


if __name__ == "__main__":
    main()
