"""CPU: the oracle (oracle/cpu_ref.py, oracle/problems_ref.py) against the committed
golden vectors (outputs of the REFERENCE solver, tests/golden/make_golden.py) and
against the reference's own literal known answers."""
import json
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import cpu_ref, problems_ref as P


def _quiet(fn, *a, **k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fn(*a, **k)


# ---- G5: literal known answers of /root/reference/tests/test_problems.py ------------
def test_jos1_known_answers():   # tests/test_problems.py:8-24
    p = P.JOS1Ref()
    x = np.array([1, 2, 3, 4, 5])
    np.testing.assert_almost_equal(p.f(x), [11, 3])
    np.testing.assert_almost_equal(p.jac_f(x), [[.4, .8, 1.2, 1.6, 2.0], [-.4, 0, .4, .8, 1.2]])


def test_jos1_l1_known_answers():   # tests/test_problems.py:27-42
    p = P.JOS1Ref(l1_ratios=[0.2, 0.1], l1_shifts=[0, 1])
    np.testing.assert_almost_equal(p.g(np.array([1, 2, 3, 4, 5])), [3, 1])
    np.testing.assert_almost_equal(p.prox_wsum_g(np.array([0.5, 0.5]), np.array([3, 4, 5, 6, 7])),
                                   [2.85, 3.85, 4.85, 5.85, 6.85])


def test_fds_known_answers():   # tests/test_problems.py:77-103
    p = P.FDSRef(n_features=5)
    x = np.array([1, 2, 3, 4, 5])
    np.testing.assert_almost_equal(p.f(x), [0.0, 75.0855369, 0.1183459])
    np.testing.assert_almost_equal(p.jac_f(x), [
        [0, 0, 0, 0, 0],
        [6.01710738, 8.01710738, 10.0171074, 12.0171074, 14.0171074],
        [-0.0613132402, -0.0360894089, -0.0149361205, -4.88417037e-03, -1.12299117e-03]])


def test_fds_constrained_known_answers():   # tests/test_problems.py:106-126
    p = P.FDSRef(n_features=5, bounds=(0, np.inf))
    np.testing.assert_almost_equal(p.g(np.ones(5)), [0, 0, 0])
    assert np.all(np.isinf(p.g(-np.ones(5))))
    np.testing.assert_almost_equal(p.prox_wsum_g(np.ones(3) / 3, np.array([-3, -1, 0, 1, 3])), [0, 0, 0, 1, 3])


def test_box_identity_prox():   # tests/test_problems.py:45-74 (SD: identity prox inside the box)
    p = P.ProblemRef(4, 2, bounds=(1e-6, np.inf))
    x = np.array([1, np.sqrt(2), np.sqrt(2), 1])
    np.testing.assert_almost_equal(p.g(x), [0, 0])
    np.testing.assert_almost_equal(p.prox_wsum_g(np.array([0.5, 0.5]), x), x)


def test_toy_lasso_optima():   # tests/test_proximal_gradient.py:99-114,151-166,204-219
    A = np.array([[-1.0], [0.0], [1.0]])
    b = np.array([-1.0, 0.0, 1.0])
    for lam, expected in [(1e-8, 1), (0.1, 0.85), (0.5, 0.25), (1, 0)]:
        base = P.LeastSquaresL1Ref(A, b, lam, scale=1 / 6)
        for m in (1, 2, 3):
            cb = base.callbacks() if m == 1 else P.stacked(base, m)
            for nest in (False, True):
                res = _quiet(cpu_ref.minimize_proximal_gradient, *cb, np.array([0.3]), nesterov=nest)
                np.testing.assert_array_almost_equal(res.x, [expected], decimal=3)


# ---- G1: toy LASSO traces ------------------------------------------------------------
def test_g1_toy_traces(golden):
    G = golden("g1_toy_lasso.npz")
    A = np.array([[-1.0], [0.0], [1.0]])
    b = np.array([-1.0, 0.0, 1.0])
    for li, lam in enumerate(G("lams")):
        base = P.LeastSquaresL1Ref(A, b, lam, scale=1 / 6)
        for m in (1, 2, 3):
            cb = base.callbacks() if m == 1 else P.stacked(base, m)
            for nest in (False, True):
                tag = f"l{li}_m{m}_{'fista' if nest else 'ista'}"
                r = _quiet(cpu_ref.minimize_proximal_gradient, *cb, np.array([0.3]), nesterov=nest, return_all=True)
                assert r.nit == int(G(f"{tag}.nit"))
                assert np.array_equal(np.stack(r.allvecs), G(f"{tag}.vecs"))
                assert np.array_equal(np.asarray(r.allerrs, float), G(f"{tag}.allerrs"))
                assert np.array_equal(np.asarray(r.allfuns, float), G(f"{tag}.allfuns"))
                assert np.array_equal(np.asarray(r.alllrs), G(f"{tag}.alllrs"))


# ---- G2: LASSO 512 x 1024 ---------------------------------------------------------------
@pytest.mark.parametrize("tag,kw", [
    ("ista", dict(nesterov=False)),
    ("fista_0_0.25", dict(nesterov=True, nesterov_ratio=(0, 0.25))),
    ("fista_0.5_0.25", dict(nesterov=True, nesterov_ratio=(0.5, 0.25))),
    ("fista_0.25_0.015625", dict(nesterov=True, nesterov_ratio=(0.25, 1 / 64))),
])
def test_g2_lasso(golden, tag, kw):
    G = golden("g2_lasso_512x1024.npz")
    A, b, lam = P.make_plasso(512, 1024, seed=0)
    assert lam == float(G("lam"))
    r = _quiet(cpu_ref.minimize_proximal_gradient, *P.LeastSquaresL1Ref(A, b, lam).callbacks(), np.zeros(1024),
               lr=1, tol=0.0, max_iter=50, return_all=True, **kw)
    assert r.nit == 50
    # same BLAS, same expressions -> equal; tolerance only guards a different OpenBLAS thread split
    np.testing.assert_allclose(np.stack([r.allvecs[k] for k in G(f"{tag}.kept")]), G(f"{tag}.vecs"),
                               rtol=0, atol=1e-13)
    np.testing.assert_allclose(r.allfuns, G(f"{tag}.allfuns"), rtol=1e-12)
    assert np.array_equal(np.asarray(r.alllrs), G(f"{tag}.alllrs"))
    assert np.array_equal(np.asarray(r.alltrials), G(f"{tag}.alltrials"))


# ---- G3: diagonal l1-quadratic -----------------------------------------------------------
@pytest.mark.parametrize("tag,kw", [
    ("fista_lr0.45", dict(lr=0.45, nesterov=True, tol=0.0, max_iter=60)),
    ("ista_lr0.45", dict(lr=0.45, nesterov=False, tol=0.0, max_iter=60)),
    ("fista_lr4_backtrack", dict(lr=4.0, nesterov=True, tol=0.0, max_iter=60)),
    ("fista_tol1e-6", dict(lr=0.45, nesterov=True, tol=1e-6, max_iter=10000)),
    ("fista_ab_0.5_0.25", dict(lr=0.45, nesterov=True, nesterov_ratio=(0.5, 0.25), tol=0.0, max_iter=60)),
])
def test_g3_diag(golden, tag, kw):
    G = golden("g3_diag_n10007.npz")
    d, c, lam = P.make_pdiag(10007, seed=1)
    r = _quiet(cpu_ref.minimize_proximal_gradient, *P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(10007),
               return_all=True, **kw)
    assert r.nit == int(G(f"{tag}.nit"))
    assert np.array_equal(np.stack([r.allvecs[k] for k in G(f"{tag}.kept")]), G(f"{tag}.vecs"))
    assert np.array_equal(np.asarray(r.allerrs, float), G(f"{tag}.allerrs"))
    assert np.array_equal(np.asarray(r.allfuns, float), G(f"{tag}.allfuns"))
    assert np.array_equal(np.asarray(r.alltrials), G(f"{tag}.alltrials"))


def test_momentum_sequence_matches_product_table():
    from zfista_amd.engine import momentum_factors

    for ratio in [(0, 0.25), (0.5, 0.25), (0.25, 1 / 64), (0.75, 0.1)]:
        ref = cpu_ref.momentum_sequence(300, ratio)
        a, st = momentum_factors(120, ratio)
        b, st = momentum_factors(180, ratio, st)
        assert np.array_equal(np.concatenate([a, b]), ref)
    assert cpu_ref.momentum_sequence(1)[0] == 0.0


# ---- G4: multi-objective -------------------------------------------------------------------
def _g4_cases():
    return {
        "jos1_n50": (lambda: P.JOS1Ref(50), dict(lr=1.0)),
        "jos1_n50_l1": (lambda: P.JOS1Ref(50, l1_ratios=np.arange(1, 3) / 50, l1_shifts=[0, 1]), dict(lr=1.0)),
        "jos1_n1000_l1": (lambda: P.JOS1Ref(1000, l1_ratios=np.arange(1, 3) / 1000, l1_shifts=[0, 1]), dict(lr=1.0)),
        "jos1_n50_box": (lambda: P.JOS1Ref(50, bounds=(-1.0, 1.5)), dict(lr=1.0)),
        "fds_n10": (lambda: P.FDSRef(10), dict(lr=0.05)),
        "fds_n10_l1": (lambda: P.FDSRef(10, l1_ratios=np.arange(1, 4) / 10, l1_shifts=[0, 1, 2]), dict(lr=0.05)),
        "fds_n100_l1": (lambda: P.FDSRef(100, l1_ratios=np.arange(1, 4) / 100, l1_shifts=[0, 1, 2]), dict(lr=1e-3)),
        "fds_n10_pos": (lambda: P.FDSRef(10, bounds=(0, np.inf)), dict(lr=0.05)),
    }


@pytest.mark.parametrize("tag", list(_g4_cases()))
def test_g4_multiobjective(golden, tag):
    G = golden("g4_multiobjective.npz")
    make, kw = _g4_cases()[tag]
    prob = make()
    x0 = G(f"{tag}.x0")
    for nest in (False, True):
        v = "fista" if nest else "ista"
        r = _quiet(cpu_ref.minimize_proximal_gradient, *prob.callbacks(), x0, nesterov=nest, tol=1e-5,
                   max_iter=12, return_all=True, **kw)
        assert r.nit == int(G(f"{tag}.{v}.nit"))
        np.testing.assert_allclose(np.stack(r.allvecs), G(f"{tag}.{v}.vecs"), rtol=0, atol=1e-12)
        np.testing.assert_allclose(np.stack(r.allfuns), G(f"{tag}.{v}.allfuns"), rtol=1e-11)
    w0 = np.ones(prob.n_objectives) / prob.n_objectives
    s = _quiet(cpu_ref.trial_multi, *prob.callbacks(), float(G(f"{tag}.sub.lr")), x0, G(f"{tag}.sub.y"), w0,
               tol=1e-12, max_iter=100000)
    np.testing.assert_allclose(s.x, G(f"{tag}.sub.x"), rtol=0, atol=1e-12)
    np.testing.assert_allclose(s.weight, G(f"{tag}.sub.weight"), rtol=0, atol=1e-9)
    np.testing.assert_allclose(s.fun, float(G(f"{tag}.sub.fun")), rtol=1e-10)


# ---- G6: result-dict shapes -------------------------------------------------------------------
def test_g6_result_shapes(capsys):
    shapes = json.load(open(os.path.join(GOLDEN, "g6_result_shapes.json")))
    A = np.array([[-1.0], [0.0], [1.0]])
    b = np.array([-1.0, 0.0, 1.0])
    prob = P.LeastSquaresL1Ref(A, b, 0.1, scale=1 / 6)
    x0 = np.array([0.3])
    f, g, jac, prox = prob.callbacks()

    def bad(x):
        raise ValueError("boom in jac_f")

    runs = {
        "success": (prob.callbacks(), {}),
        "success_return_all": (prob.callbacks(), dict(return_all=True)),
        "max_iter": (prob.callbacks(), dict(max_iter=3)),
        "deprecated": (prob.callbacks(), dict(deprecated=True)),
        "callback_exception": ((f, g, bad, prox), {}),
        "backtracking_failure": ((f, g, lambda x: -1e6 * np.ones_like(x), prox), dict(max_backtrack_iter=5, lr=1e3)),
    }
    for name, (cb, kw) in runs.items():
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            r = cpu_ref.minimize_proximal_gradient(*cb, x0, **kw)
        exp = shapes[name]
        assert sorted(k for k in r.keys() if k not in ("alllrs", "alltrials")) == exp["keys"], name
        for k in ("status", "message", "success", "nit"):
            if k in exp:
                assert r[k] == exp[k], (name, k)
        assert [[x.category.__name__, str(x.message)] for x in w] == exp["warnings"], name
        out = capsys.readouterr().out.splitlines()
        assert out[:1] == exp["stdout_ref"], name
