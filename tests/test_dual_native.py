"""CPU: the opt-in native dual solver (zfista_amd.multiobjective.solve_dual_native,
SURVEY 8f rank 1) on the ORACLE's dual function: it must reach a dual value at least as
good as the reference's SciPy calls, satisfy the KKT conditions of the simplex-constrained
problem, and do so in tens of evaluations."""
import warnings

import numpy as np
import pytest

from oracle import cpu_ref, problems_ref as P
from zfista_amd.multiobjective import _simplex_qp, solve_dual_native

CASES = {
    "jos1_n1000_l1": (lambda: P.JOS1Ref(1000, l1_ratios=np.arange(1, 3) / 1000, l1_shifts=[0, 1]), 1.0),
    "jos1_n50": (lambda: P.JOS1Ref(50), 1.0),
    "fds_n100_l1": (lambda: P.FDSRef(100, l1_ratios=np.arange(1, 4) / 100, l1_shifts=[0, 1, 2]), 1e-3),
    "fds_n10_pos": (lambda: P.FDSRef(10, bounds=(0, np.inf)), 0.05),
    "fds_n10": (lambda: P.FDSRef(10), 0.05),
    "fds_n10_small_lr": (lambda: P.FDSRef(10), 0.005),
    "fds_n1000_l1": (lambda: P.FDSRef(1000, l1_ratios=np.arange(1, 4) / 1000, l1_shifts=[0, 1, 2]), 1e-5),
}


@pytest.mark.parametrize("tag", list(CASES))
def test_native_dual_reaches_kkt_point(tag):
    make, lr = CASES[tag]
    prob = make()
    n, m = prob.n_features, prob.n_objectives
    rng = np.random.default_rng(100 + list(CASES).index(tag))
    x0 = rng.uniform(0, 2, n) if "pos" in tag else rng.uniform(-2, 2, n)
    y = x0 + 0.1 * rng.standard_normal(n)
    if "pos" in tag:
        y = np.abs(y)
    f_y, F_old, J = prob.f(y), prob.f(x0) + prob.g(x0), prob.jac_f(y)
    evals = [0]

    def dual(w):
        evals[0] += 1
        return cpu_ref.dual_value_and_grad(w, prob.g, prob.prox_wsum_g, lr, y, J, f_y, F_old)

    w, fun, nit = solve_dual_native(dual, m, None, 1e-12, 100)
    n_native = evals[0]
    assert n_native <= 80, n_native
    assert abs(w.sum() - 1) <= 1e-12 and np.all(w >= 0)
    # KKT on the simplex: the gradient is constant (= its minimum) on the support and not
    # smaller off it
    _, grad = dual(w)
    scale = 1 + np.abs(grad).max()
    support = w > 1e-9
    lo = grad[support].min()
    assert np.all(np.abs(grad[support] - lo) <= 1e-6 * scale)
    assert np.all(grad[~support] >= lo - 1e-6 * scale)
    if "n1000" not in tag:   # at lr = 1e-5 / n = 1000 trust-constr runs to max_iter (minutes)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s = cpu_ref.trial_multi(*prob.callbacks(), lr, x0, y, np.ones(m) / m, tol=1e-12, max_iter=100000)
        # minimisation: the native dual value may not be worse than SciPy's (fun = -dual)
        assert fun <= -s.fun + 1e-9 * (1 + abs(s.fun))
        x_nat = prob.prox_wsum_g(lr * w, y - lr * w @ J)
        assert np.linalg.norm(x_nat - s.x) <= 2e-5 * np.linalg.norm(s.x)


def test_simplex_qp_small_cases():
    # interior solution
    Q = np.diag([1.0, 1.0, 1.0])
    w = _simplex_qp(np.zeros(3), Q)
    np.testing.assert_allclose(w, np.ones(3) / 3, atol=1e-14)
    # vertex
    w = _simplex_qp(np.array([0.0, 5.0, 5.0]), 1e-3 * np.eye(3))
    np.testing.assert_allclose(w, [1, 0, 0], atol=1e-12)
    # edge
    w = _simplex_qp(np.array([0.0, 0.0, 9.0]), np.eye(3))
    np.testing.assert_allclose(w, [0.5, 0.5, 0], atol=1e-12)
    # brute force on a grid for a random convex instance
    rng = np.random.default_rng(0)
    B = rng.standard_normal((3, 3))
    Q = B @ B.T
    q = rng.standard_normal(3)
    w = _simplex_qp(q, Q)
    g = np.linspace(0, 1, 201)
    best = min((q @ np.array([a, b, 1 - a - b]) + 0.5 * np.array([a, b, 1 - a - b]) @ Q @ np.array([a, b, 1 - a - b]))
               for a in g for b in g if a + b <= 1)
    assert q @ w + 0.5 * w @ Q @ w <= best + 1e-9


def test_infinite_F_old_defers_to_scipy():
    prob = P.JOS1Ref(20, bounds=(-1.0, 1.5))
    x0 = np.full(20, 3.0)          # outside the box: F(x0) = inf
    y = x0.copy()

    def dual(w):
        return cpu_ref.dual_value_and_grad(w, prob.g, prob.prox_wsum_g, 1.0, y, prob.jac_f(y), prob.f(y),
                                           prob.f(x0) + prob.g(x0))

    with np.errstate(invalid="ignore"):
        assert solve_dual_native(dual, 2, None, 1e-12, 100) is None
