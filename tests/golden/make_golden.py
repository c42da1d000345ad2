#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REFERENCE solver.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):  ``python tests/golden/make_golden.py``

For every case the imported reference ``zfista.minimize_proximal_gradient``
(/root/reference/zfista/proximal_gradient.py:311) and the oracle restatement
``oracle.cpu_ref.minimize_proximal_gradient`` are run on the SAME callbacks
(``oracle/problems_ref.py`` closures; jaxopt is absent so the prox is the NumPy
``sign*max(|x|-t,0)`` restatement) and must agree exactly; the reference's
outputs are what is stored.  Nothing of the reference's source is stored - the
fixtures are inputs (seeds / literals) and expected outputs only.

Cases:
  G1  toy LASSO of tests/test_proximal_gradient.py:75-78 from x0 = 0.3,
      m = 1 and the duplicated m = 2 / m = 3 variants, ISTA + FISTA
  G2  LASSO 512x1024 (seed 0), ISTA + three momentum ratios, 50 iterations
  G3  diagonal l1-quadratic: n = 10007 full scalar traces + sampled iterates
      (lr 0.45 and a backtracking run from lr 4), n = 10^7 K = 20 scalars/samples
  G4  JOS1 (m = 2) and FDS (m = 3) traces + direct _solve_subproblem captures
  G6  result-dict shapes: success / max-iter / callback exception /
      backtracking failure / deprecated=True
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from zfista import minimize_proximal_gradient as ref_solve  # noqa: E402
from zfista.proximal_gradient import _solve_subproblem as ref_subproblem  # noqa: E402

from oracle import cpu_ref, problems_ref as P  # noqa: E402


def run_both(cb, x0, **kw):
    """Reference and oracle on the same callbacks; assert exact agreement."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with contextlib.redirect_stdout(io.StringIO()):
            r = ref_solve(*cb, x0, **kw)
            o = cpu_ref.minimize_proximal_gradient(*cb, x0, **kw)
    assert r.nit == o.nit, (r.nit, o.nit)
    assert np.array_equal(np.asarray(r.x), np.asarray(o.x)), "x differs"
    assert np.array_equal(np.asarray(r.fun), np.asarray(o.fun)), "fun differs"
    assert r.success == o.success and r.message == o.message
    if kw.get("return_all"):
        assert len(r.allvecs) == len(o.allvecs)
        for a, b in zip(r.allvecs, o.allvecs):
            assert np.array_equal(a, b)
        assert np.array_equal(np.asarray(r.allerrs), np.asarray(o.allerrs))
        assert np.array_equal(np.asarray(r.allfuns), np.asarray(o.allfuns))
    return r, o


def trace_arrays(r, o, keep=None):
    """Scalar traces (+ selected iterates) of a return_all run."""
    out = dict(
        x=np.asarray(r.x), fun=np.asarray(r.fun), nit=np.int64(r.nit),
        allerrs=np.asarray(r.allerrs, float), allfuns=np.asarray(r.allfuns, float),
        alllrs=np.asarray(o.alllrs, float), alltrials=np.asarray(o.alltrials, np.int64),
        status=np.int64(r.get("status", -1)),
    )
    if keep is not None:
        keep = [k for k in keep if k < len(r.allvecs)]
        out["kept"] = np.asarray(keep, np.int64)
        out["vecs"] = np.stack([r.allvecs[k] for k in keep])
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def g1_toy():
    A = np.array([[-1.0], [0.0], [1.0]])
    b = np.array([-1.0, 0.0, 1.0])
    x0 = np.array([0.3])
    out = {}
    for li, lam in enumerate([1e-8, 0.1, 0.5, 1.0]):
        base = P.LeastSquaresL1Ref(A, b, lam, scale=1.0 / 6.0)
        for m in (1, 2, 3):
            cb = base.callbacks() if m == 1 else P.stacked(base, m)
            for nest in (False, True):
                r, o = run_both(cb, x0, nesterov=nest, return_all=True)
                tag = f"l{li}_m{m}_{'fista' if nest else 'ista'}"
                t = trace_arrays(r, o, keep=range(len(r.allvecs)))
                for k, v in t.items():
                    out[f"{tag}.{k}"] = v
    out["lams"] = np.array([1e-8, 0.1, 0.5, 1.0])
    save("g1_toy_lasso.npz", **out)


def g2_lasso():
    A, b, lam = P.make_plasso(512, 1024, seed=0)
    prob = P.LeastSquaresL1Ref(A, b, lam, scale=0.5)
    x0 = np.zeros(1024)
    out = dict(lam=np.float64(lam))
    variants = {
        "ista": dict(nesterov=False),
        "fista_0_0.25": dict(nesterov=True, nesterov_ratio=(0, 0.25)),
        "fista_0.5_0.25": dict(nesterov=True, nesterov_ratio=(0.5, 0.25)),
        "fista_0.25_0.015625": dict(nesterov=True, nesterov_ratio=(0.25, 1 / 64)),
    }
    for tag, kw in variants.items():
        r, o = run_both(prob.callbacks(), x0, lr=1, tol=0.0, max_iter=50, return_all=True, **kw)
        t = trace_arrays(r, o, keep=list(range(0, 51, 5)))
        for k, v in t.items():
            out[f"{tag}.{k}"] = v
    save("g2_lasso_512x1024.npz", **out)


def g3_diag():
    n = 10007
    d, c, lam = P.make_pdiag(n, seed=1)
    prob = P.DiagQuadL1Ref(d, c, lam)
    x0 = np.zeros(n)
    out = {}
    runs = {
        "fista_lr0.45": dict(lr=0.45, nesterov=True, tol=0.0, max_iter=60),
        "ista_lr0.45": dict(lr=0.45, nesterov=False, tol=0.0, max_iter=60),
        "fista_lr4_backtrack": dict(lr=4.0, nesterov=True, tol=0.0, max_iter=60),
        "fista_tol1e-6": dict(lr=0.45, nesterov=True, tol=1e-6, max_iter=10000),
        "fista_ab_0.5_0.25": dict(lr=0.45, nesterov=True, nesterov_ratio=(0.5, 0.25),
                                  tol=0.0, max_iter=60),
    }
    for tag, kw in runs.items():
        r, o = run_both(prob.callbacks(), x0, return_all=True, **kw)
        t = trace_arrays(r, o, keep=[1, 2, 10, r.nit])
        for k, v in t.items():
            out[f"{tag}.{k}"] = v
    save("g3_diag_n10007.npz", **out)

    n = 10**7
    d, c, lam = P.make_pdiag(n, seed=1)
    prob = P.DiagQuadL1Ref(d, c, lam)
    r, o = run_both(prob.callbacks(), np.zeros(n), lr=0.45, nesterov=True, tol=0.0,
                    max_iter=20, return_all=True)
    save(
        "g3_diag_n1e7.npz",
        allerrs=np.asarray(r.allerrs), allfuns=np.asarray(r.allfuns),
        alllrs=np.asarray(o.alllrs), alltrials=np.asarray(o.alltrials, np.int64),
        x_norm=np.float64(np.linalg.norm(r.x)), x_sample=r.x[::100003].copy(),
        x5_sample=r.allvecs[5][::100003].copy(), fun=np.float64(r.fun),
    )


def g4_multi():
    out = {}
    rng = np.random.default_rng(7)
    cases = {
        "jos1_n50": (P.JOS1Ref(50), dict(lr=1.0)),
        "jos1_n50_l1": (P.JOS1Ref(50, l1_ratios=np.arange(1, 3) / 50, l1_shifts=[0, 1]), dict(lr=1.0)),
        "jos1_n1000_l1": (P.JOS1Ref(1000, l1_ratios=np.arange(1, 3) / 1000, l1_shifts=[0, 1]), dict(lr=1.0)),
        "jos1_n50_box": (P.JOS1Ref(50, bounds=(-1.0, 1.5)), dict(lr=1.0)),
        "fds_n10": (P.FDSRef(10), dict(lr=0.05)),
        "fds_n10_l1": (P.FDSRef(10, l1_ratios=np.arange(1, 4) / 10, l1_shifts=[0, 1, 2]), dict(lr=0.05)),
        "fds_n100_l1": (P.FDSRef(100, l1_ratios=np.arange(1, 4) / 100, l1_shifts=[0, 1, 2]), dict(lr=1e-3)),
        "fds_n10_pos": (P.FDSRef(10, bounds=(0, np.inf)), dict(lr=0.05)),
    }
    for tag, (prob, kw) in cases.items():
        n = prob.n_features
        x0 = rng.uniform(-2, 2, n) if "pos" not in tag else rng.uniform(0, 2, n)
        out[f"{tag}.x0"] = x0
        for nest in (False, True):
            r, o = run_both(prob.callbacks(), x0, nesterov=nest, tol=1e-5, max_iter=12,
                            return_all=True, **kw)
            t = trace_arrays(r, o, keep=range(len(r.allvecs)))
            for k, v in t.items():
                out[f"{tag}.{'fista' if nest else 'ista'}.{k}"] = v
        # direct subproblem capture at a fixed (lr, x_old, y)
        y = x0 + 0.1 * rng.standard_normal(n)
        if "pos" in tag:
            y = np.abs(y)
        lr = kw["lr"]
        w0 = np.ones(prob.n_objectives) / prob.n_objectives
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s = ref_subproblem(*prob.callbacks(), lr, x0, y, w0, tol=1e-12, max_iter=100000)
            so = cpu_ref.trial_multi(*prob.callbacks(), lr, x0, y, w0, tol=1e-12, max_iter=100000)
        assert np.array_equal(s.x, so.x) and s.fun == so.fun and s.nit == so.nit
        out[f"{tag}.sub.y"] = y
        out[f"{tag}.sub.lr"] = np.float64(lr)
        out[f"{tag}.sub.x"] = s.x
        out[f"{tag}.sub.weight"] = s.weight
        out[f"{tag}.sub.fun"] = np.float64(s.fun)
        out[f"{tag}.sub.nit"] = np.int64(s.nit)
    save("g4_multiobjective.npz", **out)


def g7_library():
    """The reference SOLVER (imported) on the remaining problem families of zfista/problems.py,
    whose callbacks are the oracle's restatements (zfista.problems itself needs jax)."""
    out = {}
    rng = np.random.default_rng(11)
    cases = {
        "sd": (P.SDRef(), rng.uniform(1.5, 2.5, 4), dict(lr=0.5)),
        "zdt1_n30": (P.ZDT1Ref(30), rng.uniform(0.2, 0.8, 30), dict(lr=0.1)),
        "toi4": (P.TOI4Ref(), rng.uniform(-2, 2, 4), dict(lr=0.5)),
        "toi4_l1": (P.TOI4Ref(l1_ratios=[0.1, 0.2], l1_shifts=[0.0, 0.5]), rng.uniform(-2, 2, 4), dict(lr=0.5)),
        "tridia": (P.TRIDIARef(), rng.uniform(-1, 1, 3), dict(lr=0.05)),
        "tridia_l1_box": (P.TRIDIARef(l1_ratios=[0.1, 0.05, 0.02], l1_shifts=[0.0, 0.1, 0.2], bounds=(-0.5, 0.9)),
                          rng.uniform(-0.4, 0.8, 3), dict(lr=0.05)),
        "lfr1": (P.LinearFunctionRank1Ref(), rng.uniform(-0.1, 0.1, 10), dict(lr=1e-4)),
    }
    for tag, (prob, x0, kw) in cases.items():
        out[f"{tag}.x0"] = x0
        for nest in (False, True):
            r, o = run_both(prob.callbacks(), x0, nesterov=nest, tol=1e-6, max_iter=15, return_all=True, **kw)
            t = trace_arrays(r, o, keep=range(len(r.allvecs)))
            for k, v in t.items():
                out[f"{tag}.{'fista' if nest else 'ista'}.{k}"] = v
    save("g7_problem_library.npz", **out)


def g6_shapes():
    A = np.array([[-1.0], [0.0], [1.0]])
    b = np.array([-1.0, 0.0, 1.0])
    prob = P.LeastSquaresL1Ref(A, b, 0.1, scale=1.0 / 6.0)
    x0 = np.array([0.3])
    shapes = {}

    def describe(r, warned):
        d = {"keys": sorted(r.keys()), "warnings": warned}
        for k in ("status", "message", "success", "nit"):
            if k in r:
                v = r[k]
                d[k] = v if isinstance(v, (str, bool)) else int(v)
        for k in ("allvecs", "allfuns", "allerrs"):
            if k in r:
                d[k + "_is_none"] = r[k] is None
        return d

    def run(name, cb, **kw):
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                r = ref_solve(*cb, x0, **kw)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    o = cpu_ref.minimize_proximal_gradient(*cb, x0, **kw)
        shapes[name] = describe(r, [(x.category.__name__, str(x.message)) for x in w])
        shapes[name]["stdout_ref"] = buf.getvalue().splitlines()[:1]
        assert sorted(r.keys()) == sorted(k for k in o.keys() if k not in ("alllrs", "alltrials")), name
        for k in ("status", "message", "success", "nit"):
            if k in r:
                assert r[k] == o[k], (name, k)

    run("success", prob.callbacks())
    run("success_return_all", prob.callbacks(), return_all=True)
    run("max_iter", prob.callbacks(), max_iter=3)
    run("deprecated", prob.callbacks(), deprecated=True)

    def bad_jac(x):
        raise ValueError("boom in jac_f")

    f, g, _, prox = prob.callbacks()
    run("callback_exception", (f, g, bad_jac, prox))

    # never-accepted line search: model is made hopeless by a lying jac_f
    run("backtracking_failure", (f, g, lambda x: -1e6 * np.ones_like(x), prox),
        max_backtrack_iter=5, lr=1e3)

    with open(os.path.join(HERE, "g6_result_shapes.json"), "w") as fh:
        json.dump(shapes, fh, indent=1, sort_keys=True)
    print("wrote g6_result_shapes.json")


if __name__ == "__main__":
    only = sys.argv[1:]
    todo = dict(g1=g1_toy, g2=g2_lasso, g3=g3_diag, g4=g4_multi, g6=g6_shapes, g7=g7_library)
    for k, fn in todo.items():
        if not only or k in only:
            fn()
