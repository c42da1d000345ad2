"""GPU: the experiment harness (tools/harness.py: the design of benchmarks/benchmark.py:303-374 - three
solver variants per random start, metric tables of zfista/metrics.py:103-199) and the replica launcher
behind it (zfista_amd/replicas.py: the reference's joblib sweeps as one worker process per GPU)."""
import importlib.util
import json
import os
import warnings

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _harness():
    spec = importlib.util.spec_from_file_location("zf_harness", os.path.join(ROOT, "tools", "harness.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_harness_jos1_sweep_and_metric_tables(tmp_path):
    H = _harness()
    out = tmp_path / "h.json"
    rep = H.main(["--samples", "6", "--problems", "JOS1", "--max-n", "10", "--out", str(out)])
    assert json.load(open(out)).keys() == rep.keys() and len(rep) == 4   # JOS1 n = 5, 10, each with its l1 twin
    for name, r in rep.items():
        m = r["metrics"]
        assert set(m["Avg iterations"]) == {"Normal", "Accelerated", "Accelerated (deprecated test)"}
        # every start converges (tol 1e-5)
        assert all(v is not None and v > 0 for v in m["Avg iterations"].values()), name
        assert all(0.0 <= v <= 1.0 for v in m["Purity"].values()) and all(v == 0.0 for v in m["Error rate"].values())
        assert all(v >= 1.0 for v in r["ratios"]["Avg iterations"].values())   # value / best
        assert all(v is None or v > 0 for v in r["iterations_per_second"].values())
    # --- the same design through the ORACLE: the same six starts per problem (the harness draws them from
    # default_rng(seed) in problem order), the three variants of benchmarks/benchmark.py:303-374 solved by
    # oracle.cpu_ref on oracle.problems_ref, tabulated by the same zfista_amd.metrics (zfista/metrics.py:103-199)
    from oracle import cpu_ref, problems_ref as P
    from zfista_amd.metrics import calculate_metrics

    rng = np.random.default_rng(0)
    variants = {"Normal": {}, "Accelerated": dict(nesterov=True),
                "Accelerated (deprecated test)": dict(nesterov=True, deprecated=True)}
    for prob in H.build_problems(["JOS1"], 10):
        n, m = prob.n_features, prob.n_objectives
        starts = rng.uniform(-2, 4, size=(6, n))
        ref = P.JOS1Ref(n) if prob.l1_ratios is None else P.JOS1Ref(n, l1_ratios=(np.arange(m) + 1) / n,
                                                                     l1_shifts=np.arange(m))
        res, res_perm = {}, {}
        for label, kw in variants.items():
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res[label] = [cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, return_all=True,
                                                                 max_iter=100000000, tol_internal=1e-11, **kw)
                              for x0 in starts]
                # the oracle's OWN spread (G10's rule): the same starts with their features permuted - JOS1 is
                # symmetric in its features, so this is the same problem with every sum in another order
                res_perm[label] = [cpu_ref.minimize_proximal_gradient(
                    *ref.callbacks(), x0[np.random.default_rng(100 + k).permutation(n)], max_iter=100000000,
                    tol_internal=1e-11, **kw) for k, x0 in enumerate(starts)]
        metrics, ratios = calculate_metrics(*res.items())
        got = rep[prob.name]
        for label in variants:
            # Round 4: no flat 15 % any more.  The average iteration count is held to 10 x the spread the oracle itself
            # shows between the two summation orders - measured per problem and variant, "Normal" included (zero for
            # most; JOS1 n = 10 + l1: 47.0 vs 47.67 "Normal", 31.17 vs 31.5 "Accelerated") - and never tighter than
            # one iteration on one of the six starts (a stop at err < tol = 1e-5 is a knife edge on a trajectory that
            # depends on the last bits of Brent's dual solves).
            want_it = metrics["Avg iterations"][label]
            spread = abs(want_it - float(np.mean([r.nit for r in res_perm[label]])))
            tol_it = max(10.0 * spread, 1.0 / len(starts) + 1e-9)
            assert abs(got["metrics"]["Avg iterations"][label] - want_it) <= tol_it, \
                (prob.name, label, got["metrics"]["Avg iterations"][label], want_it, spread)
        # Hypervolume is a continuous function of the objective vectors: equal to 1e-3.  Purity / Gamma / Delta are
        # not - they count which variant's point DOMINATES when two variants reach the same Pareto point from the
        # same start up to ~1e-8 (the accuracy of Brent's dual solves), so the oracle's and the engine's tables may
        # differ there exactly like two runs of the reference with another summation order do (measured: Gamma of
        # JOS1 n = 10 "Normal" 1.45 vs 2.25): only their ranges are checked
        for key in ("Hypervolume", "Error rate"):   # (the fronts are those of solves stopped at tol = 1e-5: 1e-3 of the volume)
            for label in variants:
                assert got["metrics"][key][label] == pytest.approx(float(metrics[key][label]), rel=1e-3, abs=1e-9), \
                    (prob.name, key, label)
        for label in variants:
            assert 0.0 <= got["metrics"]["Purity"][label] <= 1.0 and got["metrics"]["Gamma"][label] >= 0.0
    # the same sweep as replicas (2 worker processes sharing this GPU): identical solves, identical tables
    rep2 = H.main(["--samples", "6", "--problems", "JOS1", "--max-n", "10", "--workers", "2", "--out", str(out)])
    for name in rep:
        assert rep2[name]["metrics"]["Avg iterations"] == rep[name]["metrics"]["Avg iterations"], name
        for key in ("Hypervolume", "Purity", "Gamma", "Delta", "Avg iterations"):
            assert rep2[name]["ratios"][key] == rep[name]["ratios"][key], (name, key)


def _make_closure_problem():
    """The reference's toy LASSO closures (tests/test_proximal_gradient.py:75-97): opaque callbacks."""
    A = np.array([[-1.0], [0.0], [1.0]])
    b = np.array([-1.0, 0.0, 1.0])

    def f(x):
        return np.linalg.norm(A @ x - b) ** 2 / 6

    def g(x):
        return 0.1 * np.linalg.norm(x, ord=1)

    def jac_f(x):
        return A.T @ (A @ x - b) / 3

    def prox(weight, x):
        return np.sign(x) * np.maximum(np.abs(x) - 0.1 * weight, 0)

    return f, g, jac_f, prox


def test_replicas_of_opaque_callbacks():
    """Generic Python callbacks: "replicas only" (SURVEY 8e).  Three workers, five starts."""
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.replicas import solve_replicas

    starts = [np.array([v]) for v in (0.3, -1.0, 2.0, 0.0, 0.7)]
    res = solve_replicas(_make_closure_problem, starts, workers=3, nesterov=True, return_all=True)
    assert len(res) == 5
    for x0, r in zip(starts, res):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            here = minimize_proximal_gradient(*_make_closure_problem(), x0, nesterov=True, return_all=True)
        assert r.nit == here.nit and np.array_equal(r.x, here.x) and r.success
        np.testing.assert_array_almost_equal(r.x, [0.85], decimal=3)
        assert len(r.allvecs) == r.nit + 1
