"""GPU parity: dense least-squares + l1 (K14) through the device-resident path
and through the generic (opaque-callback) path, against golden vectors from the
reference solver and the oracle run live.  Tolerance 1e-10 relative (north_star);
GEMV reduction order differs from OpenBLAS, so no bit-exactness is claimed here."""
import warnings

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10

VARIANTS = {
    "ista": dict(nesterov=False),
    "fista_0_0.25": dict(nesterov=True, nesterov_ratio=(0, 0.25)),
    "fista_0.5_0.25": dict(nesterov=True, nesterov_ratio=(0.5, 0.25)),
    "fista_0.25_0.015625": dict(nesterov=True, nesterov_ratio=(0.25, 1 / 64)),
}


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_lasso_512x1024_golden(tag, golden):
    """BASELINE cfg1: backtracks 1 -> 2^-11 in the first iteration."""
    from oracle import problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import LeastSquaresL1

    G = golden("g2_lasso_512x1024.npz")
    A, b, lam = P.make_plasso(512, 1024, seed=0)
    assert lam == float(G("lam"))
    prob = LeastSquaresL1(A, b, lam, scale=0.5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*prob.callbacks(), np.zeros(1024), lr=1, tol=0.0, max_iter=50,
                                         return_all=True, **VARIANTS[tag])
    assert res.nit == 50
    np.testing.assert_allclose(res.allfuns, G(f"{tag}.allfuns"), rtol=TOL)
    np.testing.assert_allclose(res.allerrs, G(f"{tag}.allerrs"), rtol=1e-9, atol=1e-14)
    for k, v in zip(G(f"{tag}.kept"), G(f"{tag}.vecs")):
        assert rel_err(res.allvecs[k], v) <= TOL, k


def test_lasso_lr_sequence(golden):
    from oracle import problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.problems import LeastSquaresL1
    from zfista_amd.proximal_gradient import NativeRun

    G = golden("g2_lasso_512x1024.npz")
    A, b, lam = P.make_plasso(512, 1024, seed=0)
    prob = LeastSquaresL1(A, b, lam)
    o = dict(lr=1, tol=0.0, tol_internal=1e-12, max_iter=50, max_backtrack_iter=100, decay_rate=0.5,
             nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False)
    run = NativeRun(prob, np.zeros(1024), o)
    rows = []
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(16))
    rows = np.concatenate(rows)
    assert np.array_equal(rows[:, _lib.TR_LR], G("fista_0_0.25.alllrs"))
    assert np.array_equal(rows[:, _lib.TR_TRIALS].astype(np.int64), G("fista_0_0.25.alltrials"))
    run.solver.close()


@pytest.mark.parametrize("shape", [(3, 1), (7, 5), (64, 33), (33, 64), (129, 1000), (1000, 130),
                                   (7, 32), (16, 128), (100, 96), (257, 2048)])
def test_lasso_vs_oracle_shapes(shape):
    """Odd / even n (scalar and 16-B VALU kernels), n % 32 == 0 (MFMA A^T r kernel, with
    row counts that are / are not multiples of its 16-row step), tall and wide A."""
    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import LeastSquaresL1

    m, n = shape
    A, b, lam = P.make_plasso(m, n, seed=4, n_informative=max(1, n // 4))
    prob, ref = LeastSquaresL1(A, b, lam, scale=0.5), P.LeastSquaresL1Ref(A, b, lam, scale=0.5)
    kw = dict(lr=1.0, nesterov=True, tol=1e-7, max_iter=60, return_all=True)
    x0 = np.zeros(n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*prob.callbacks(), x0, **kw)
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **kw)
    assert res.nit == exp.nit
    for a, e in zip(res.allvecs, exp.allvecs):
        assert rel_err(a, e) <= TOL
    np.testing.assert_allclose(res.allfuns, exp.allfuns, rtol=TOL)


def test_lasso_2048x8192_vs_oracle():
    """A mid-size dense problem (128 MiB A: several row slices, many column panels) against
    the oracle on the host; lr = 0.9/L so no trial is marginal."""
    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import LeastSquaresL1

    m, n = 2048, 8192
    A, b, lam = P.make_plasso(m, n, seed=9)
    L = np.linalg.norm(A, 2) ** 2
    kw = dict(lr=0.9 / L, nesterov=True, tol=0.0, max_iter=25, return_all=True)
    prob, ref = LeastSquaresL1(A, b, lam), P.LeastSquaresL1Ref(A, b, lam)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*prob.callbacks(), np.zeros(n), **kw)
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), np.zeros(n), **kw)
    assert res.nit == exp.nit == 25
    for k in (1, 5, 25):
        assert rel_err(res.allvecs[k], exp.allvecs[k]) <= TOL
    np.testing.assert_allclose(res.allfuns, exp.allfuns, rtol=TOL)
    assert np.array_equal(np.asarray(exp.alltrials), np.ones(25, dtype=int))


def test_lasso_cfg3_full_size_properties():
    """BASELINE cfg3 size (A 16384 x 65536 fp64 = 8 GiB; an oracle run does not fit the time
    budget): size-independent properties.  (1) the operator's f and grad f agree with an
    independent fp64 evaluation (torch / rocBLAS GEMV) to 1e-12; (2) with lr = 0.9/L every
    trial is accepted and F decreases monotonically under ISTA; (3) the FISTA iterate after K
    steps equals a torch fp64 restatement of the same recursion to 1e-10."""
    import torch

    from zfista_amd import _lib
    from zfista_amd.engine import momentum_factors
    from zfista_amd.problems import LeastSquaresL1
    from zfista_amd.proximal_gradient import NativeRun

    m, n = 16384, 65536
    gen = torch.Generator(device="cuda").manual_seed(3)
    A = torch.randn(m, n, dtype=torch.float64, device="cuda", generator=gen)
    xt = torch.zeros(n, dtype=torch.float64, device="cuda")
    xt[:20] = torch.randn(20, dtype=torch.float64, device="cuda", generator=gen)
    b = A @ xt + 0.01 * torch.randn(m, dtype=torch.float64, device="cuda", generator=gen)
    lam = 0.1 * float(torch.max(torch.abs(A.T @ b)))
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
    for _ in range(15):
        v = A.T @ (A @ v)
        v /= torch.linalg.norm(v)
    L = float(torch.linalg.norm(A @ v)) ** 2
    lr = 0.9 / L
    prob = LeastSquaresL1(A, b, lam)
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen) * 1e-3
    r = A @ x - b
    np.testing.assert_allclose(prob.f(x.cpu().numpy()), 0.5 * float(r @ r), rtol=1e-12)
    g_ref = (A.T @ r).cpu().numpy()
    g_got = prob.jac_f(x.cpu().numpy())
    assert np.linalg.norm(g_got - g_ref) <= 1e-12 * np.linalg.norm(g_ref)

    o = dict(lr=lr, tol=0.0, tol_internal=1e-12, max_iter=8, max_backtrack_iter=100, decay_rate=0.5,
             nesterov=False, nesterov_ratio=(0, 0.25), deprecated=False)
    run = NativeRun(prob, torch.zeros(n, dtype=torch.float64, device="cuda"), o)
    rows = run.advance(8)
    assert len(rows) == 8 and np.all(rows[:, _lib.TR_TRIALS] == 1)
    F = np.concatenate([[run.F0], rows[:, _lib.TR_F]])
    assert np.all(np.diff(F) < 0)
    run.solver.close()

    K = 6
    o.update(nesterov=True, max_iter=K)
    run = NativeRun(prob, torch.zeros(n, dtype=torch.float64, device="cuda"), o)
    rows = run.advance(K)
    xK = torch.from_numpy(run.solver.get_x()).cuda()
    run.solver.close()
    betas = np.concatenate([[0.0], momentum_factors(K, (0, 0.25))[0]])
    xk = torch.zeros(n, dtype=torch.float64, device="cuda")
    xo = xk.clone()
    for k in range(K):
        y = xk + betas[k] * (xk - xo)
        grad = A.T @ (A @ y - b)
        u = y - lr * grad
        xn = torch.sign(u) * torch.clamp(torch.abs(u) - lam * lr, min=0.0)
        xo, xk = xk, xn
    assert float(torch.linalg.norm(xK - xk) / torch.linalg.norm(xk)) <= TOL


def test_lasso_operator_callables():
    """f / g / jac_f / prox_wsum_g as plain callables (GPU-evaluated) match the oracle."""
    from oracle import problems_ref as P
    from zfista_amd.problems import DiagQuadL1, LeastSquaresL1

    A, b, lam = P.make_plasso(200, 301, seed=2)
    prob, ref = LeastSquaresL1(A, b, lam, scale=1 / 6), P.LeastSquaresL1Ref(A, b, lam, scale=1 / 6)
    x = np.random.default_rng(0).standard_normal(301)
    np.testing.assert_allclose(prob.f(x), ref.f(x), rtol=1e-13)
    np.testing.assert_allclose(prob.g(x), ref.g(x), rtol=1e-13)
    np.testing.assert_allclose(prob.jac_f(x), ref.jac_f(x), rtol=1e-11, atol=1e-11)
    assert np.array_equal(prob.prox_wsum_g(0.3, x), ref.prox_wsum_g(0.3, x))
    d, c, lam = P.make_pdiag(1001, seed=2)
    prob, ref = DiagQuadL1(d, c, lam), P.DiagQuadL1Ref(d, c, lam)
    x = np.random.default_rng(1).standard_normal(1001)
    np.testing.assert_allclose(prob.f(x), ref.f(x), rtol=1e-13)
    np.testing.assert_allclose(prob.g(x), ref.g(x), rtol=1e-13)
    assert np.array_equal(prob.jac_f(x), ref.jac_f(x))
    assert np.array_equal(prob.prox_wsum_g(0.3, x), ref.prox_wsum_g(0.3, x))


# ---- the reference's own solver tests, through the generic (opaque callback) path ----
def _toy(l1_ratio):
    """tests/test_proximal_gradient.py:75-97 with a NumPy soft-threshold (jaxopt absent)."""
    A = np.array([[-1.0], [0.0], [1.0]])
    b = np.array([-1.0, 0.0, 1.0])

    def f(x):
        return np.linalg.norm(A @ x - b) ** 2 / 6

    def g(x):
        return l1_ratio * np.linalg.norm(x, ord=1)

    def jac_f(x):
        return A.T @ (A @ x - b) / 3

    def prox_wsum_g(weight, x):
        return np.sign(x) * np.maximum(np.abs(x) - l1_ratio * weight, 0)

    return f, g, jac_f, prox_wsum_g


def test_minimize_proximal_gradient_lasso_toy():
    from zfista_amd import minimize_proximal_gradient

    x0 = np.random.random(1)
    for l1_ratio, expected in [(1e-8, 1), (0.1, 0.85), (0.5, 0.25), (1, 0)]:
        cb = _toy(l1_ratio)
        res = minimize_proximal_gradient(*cb, x0)
        res_nesterov = minimize_proximal_gradient(*cb, x0, nesterov=True)
        np.testing.assert_array_almost_equal(res.x, [expected], decimal=3)
        np.testing.assert_array_almost_equal(res_nesterov.x, [expected], decimal=3)


def test_minimize_proximal_gradient_lasso_zero_and_return_all():
    from zfista_amd import minimize_proximal_gradient

    A = np.zeros((3, 1))
    b = np.zeros(3)
    l1_ratio = 0.1
    cb = (lambda x: np.linalg.norm(A @ x - b) ** 2 / 6, lambda x: l1_ratio * np.linalg.norm(x, ord=1),
          lambda x: A.T @ (A @ x - b) / 3,
          lambda w, x: np.sign(x) * np.maximum(np.abs(x) - l1_ratio * w, 0))
    x0 = np.random.random(1)
    res = minimize_proximal_gradient(*cb, x0)
    res_nesterov = minimize_proximal_gradient(*cb, x0, nesterov=True)
    np.testing.assert_array_almost_equal(res.x, [0], decimal=3)
    np.testing.assert_array_almost_equal(res_nesterov.x, [0], decimal=3)
    res = minimize_proximal_gradient(*cb, x0, return_all=True)
    assert "allvecs" in res and "allerrs" in res


def test_generic_path_golden_toy(golden):
    """G1: full traces of the toy LASSO from x0 = 0.3 (m = 1), generic path."""
    from zfista_amd import minimize_proximal_gradient

    G = golden("g1_toy_lasso.npz")
    for li, lam in enumerate(G("lams")):
        for nest in (False, True):
            tag = f"l{li}_m1_{'fista' if nest else 'ista'}"
            res = minimize_proximal_gradient(*_toy(float(lam)), np.array([0.3]), nesterov=nest, return_all=True)
            assert res.nit == int(G(f"{tag}.nit")), tag
            np.testing.assert_allclose(np.concatenate(res.allvecs), G(f"{tag}.vecs").ravel(), rtol=TOL, atol=1e-15)
            np.testing.assert_allclose(res.allerrs, G(f"{tag}.allerrs"), rtol=1e-9, atol=1e-15)


@pytest.mark.parametrize("kw", [dict(), dict(nesterov=True), dict(return_all=True), dict(nesterov=True, max_iter=7),
                                dict(lr=8.0, nesterov=True)])
def test_generic_path_calls_the_callbacks_as_often_as_the_reference(kw):
    """Opaque callbacks are called where and as often as the reference calls them (proximal_gradient.py:140-142,
    :279, :295, :466, :472, :523, :547): the same counting closures through the oracle (pinned to the imported
    reference, G1) and through the product must count the same - g(x0) before the loop only with return_all."""
    from oracle import cpu_ref
    from zfista_amd import minimize_proximal_gradient

    def counted():
        f, g, jac_f, prox = _toy(0.1)
        n = dict(f=0, g=0, jac_f=0, prox=0)

        def wrap(name, fn):
            def inner(*a):
                n[name] += 1
                return fn(*a)
            return inner
        return n, (wrap("f", f), wrap("g", g), wrap("jac_f", jac_f), wrap("prox", prox))

    n_ref, cb_ref = counted()
    n_got, cb_got = counted()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*cb_ref, np.array([0.3]), **kw)
        res = minimize_proximal_gradient(*cb_got, np.array([0.3]), **kw)
    assert res.nit == exp.nit and n_got == n_ref, (n_got, n_ref)
