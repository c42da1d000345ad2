"""GPU parity: device-resident fused path (P-diag) vs the oracle and the golden
vectors produced by the reference solver.  Calls go through the C ABI
(libzfista_hip.so) via zfista_amd.  Tolerance: 1e-10 relative l2 on every
stored iterate (BASELINE.json north_star); the element arithmetic is built to
round like NumPy, so iterates are in fact expected to be bit-identical and the
test records that too."""
import warnings

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-10

RUNS = {
    "fista_lr0.45": dict(lr=0.45, nesterov=True, tol=0.0, max_iter=60),
    "ista_lr0.45": dict(lr=0.45, nesterov=False, tol=0.0, max_iter=60),
    "fista_lr4_backtrack": dict(lr=4.0, nesterov=True, tol=0.0, max_iter=60),
    "fista_tol1e-6": dict(lr=0.45, nesterov=True, tol=1e-6, max_iter=10000),
    "fista_ab_0.5_0.25": dict(lr=0.45, nesterov=True, nesterov_ratio=(0.5, 0.25), tol=0.0, max_iter=60),
}


def _problem(n, seed=1):
    from oracle import problems_ref as P
    from zfista_amd.problems import DiagQuadL1

    d, c, lam = P.make_pdiag(n, seed=seed)
    return DiagQuadL1(d, c, lam), P.DiagQuadL1Ref(d, c, lam)


@pytest.mark.parametrize("tag", list(RUNS))
def test_diag_golden_traces(tag, golden):
    from zfista_amd import minimize_proximal_gradient

    G = golden("g3_diag_n10007.npz")
    prob, _ = _problem(10007)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*prob.callbacks(), np.zeros(10007), return_all=True, **RUNS[tag])
    assert res.nit == int(G(f"{tag}.nit"))
    assert len(res.allerrs) == res.nit and len(res.allvecs) == res.nit + 1
    np.testing.assert_allclose(res.allerrs, G(f"{tag}.allerrs"), rtol=TOL, atol=0)
    np.testing.assert_allclose(res.allfuns, G(f"{tag}.allfuns"), rtol=TOL, atol=0)
    for k, v in zip(G(f"{tag}.kept"), G(f"{tag}.vecs")):
        assert rel_err(res.allvecs[k], v) <= TOL
        assert np.array_equal(res.allvecs[k], v), "element arithmetic is expected to round like NumPy"
    assert rel_err(res.x, G(f"{tag}.x")) <= TOL
    np.testing.assert_allclose(res.fun, G(f"{tag}.fun"), rtol=TOL)
    assert res.status == int(G(f"{tag}.status"))


@pytest.mark.parametrize("tag", list(RUNS))
def test_diag_lr_and_trial_sequence(tag, golden):
    """Per-iteration learning rate and line-search trial counts (branch parity)."""
    from zfista_amd.proximal_gradient import NativeRun
    from zfista_amd import _lib

    G = golden("g3_diag_n10007.npz")
    prob, _ = _problem(10007)
    o = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_backtrack_iter=100, decay_rate=0.5,
             nesterov=False, nesterov_ratio=(0, 0.25), deprecated=False)
    o.update(RUNS[tag])
    run = NativeRun(prob, np.zeros(10007), o)
    rows = []
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(37))
    rows = np.concatenate(rows)
    assert np.array_equal(rows[:, _lib.TR_LR], G(f"{tag}.alllrs"))
    assert np.array_equal(rows[:, _lib.TR_TRIALS].astype(np.int64), G(f"{tag}.alltrials"))
    run.solver.close()


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 511, 4097, 100003])
@pytest.mark.parametrize("nesterov", [False, True])
def test_diag_vs_oracle_ragged_sizes(n, nesterov):
    """Odd / tiny / ragged lengths (vector tails) against the oracle run live."""
    from oracle import cpu_ref
    from zfista_amd import minimize_proximal_gradient

    prob, ref = _problem(n, seed=11)
    x0 = np.random.default_rng(5).standard_normal(n)
    kw = dict(lr=3.0, nesterov=nesterov, tol=1e-9, max_iter=40, return_all=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*prob.callbacks(), x0, **kw)
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **kw)
    assert res.nit == exp.nit and res.success == exp.success and res.message == exp.message
    for a, b in zip(res.allvecs, exp.allvecs):
        assert rel_err(a, b) <= TOL
    np.testing.assert_allclose(res.allerrs, exp.allerrs, rtol=TOL, atol=1e-300)
    np.testing.assert_allclose(res.allfuns, exp.allfuns, rtol=TOL)


def test_diag_box_and_deprecated():
    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import DiagQuadL1

    n = 5001
    d, c, lam = P.make_pdiag(n, seed=3)
    lo, hi = -0.25, 0.4

    class BoxRef(P.DiagQuadL1Ref):
        def g(self, x):
            if (x < lo).any() or (x > hi).any():
                return np.inf
            return super().g(x)

        def prox_wsum_g(self, w, x):
            return P.clip_box(super().prox_wsum_g(w, x), lo, hi)

    ref = BoxRef(d, c, lam)
    prob = DiagQuadL1(d, c, lam, bounds=(lo, hi))
    x0 = np.zeros(n)
    for kw in (dict(nesterov=True), dict(nesterov=True, deprecated=True), dict(decay_rate=1, lr=0.3)):
        kw = dict(lr=2.0, tol=1e-8, max_iter=50, return_all=True) | kw
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = minimize_proximal_gradient(*prob.callbacks(), x0, **kw)
            exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, **kw)
        assert res.nit == exp.nit
        assert rel_err(res.x, exp.x) <= TOL
        assert res.x.min() >= lo and res.x.max() <= hi
        np.testing.assert_allclose(res.allfuns, exp.allfuns, rtol=TOL)


def test_diag_n1e7_golden(golden):
    """BASELINE cfg2 size: scalars and strided samples of the reference run."""
    from zfista_amd import minimize_proximal_gradient

    G = golden("g3_diag_n1e7.npz")
    n = 10**7
    prob, _ = _problem(n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # chunked run (no return_all: the iterates stay in HBM)
        res = minimize_proximal_gradient(*prob.callbacks(), np.zeros(n), lr=0.45, nesterov=True, tol=0.0,
                                         max_iter=20)
    assert res.nit == 20
    assert rel_err(res.x[::100003], G("x_sample")) <= TOL
    np.testing.assert_allclose(np.linalg.norm(res.x), G("x_norm"), rtol=TOL)
    np.testing.assert_allclose(res.fun, G("fun"), rtol=TOL)
    # scalar traces through the chunked driver
    from zfista_amd.proximal_gradient import NativeRun
    from zfista_amd import _lib

    o = dict(lr=0.45, tol=0.0, tol_internal=1e-12, max_iter=20, max_backtrack_iter=100, decay_rate=0.5,
             nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False)
    run = NativeRun(prob, np.zeros(n), o)
    rows = run.advance(20)
    np.testing.assert_allclose(rows[:, _lib.TR_ERR], G("allerrs"), rtol=TOL)
    np.testing.assert_allclose(rows[:, _lib.TR_F], G("allfuns")[1:], rtol=TOL)
    assert np.array_equal(rows[:, _lib.TR_LR], G("alllrs"))
    run.solver.close()


def test_diag_n1e8_properties():
    """Headline size n = 10^8 (no oracle run fits the time budget): size-independent
    properties.  (1) the closed-form minimiser x* = ST(c, lam/d) is a fixed point:
    one iteration from x* moves nothing and terminates; (2) from 0 the iterates
    approach x* and F decreases to F(x*) from above; (3) a strided sample of the
    iterate equals the oracle's elementwise recursion on that sample."""
    import torch
    from oracle import problems_ref as P
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun
    from zfista_amd import _lib, minimize_proximal_gradient

    n = 10**8
    gen = torch.Generator(device="cuda").manual_seed(1)
    d = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 1.5 + 0.5
    c = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
    lam = 0.1
    prob = DiagQuadL1(d, c, lam)
    xstar = torch.sign(c) * torch.clamp(torch.abs(c) - lam / d, min=0.0)
    o = dict(lr=0.45, tol=1e-9, tol_internal=1e-12, max_iter=5, max_backtrack_iter=100, decay_rate=0.5,
             nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False)
    run = NativeRun(prob, xstar, o)
    rows = run.advance(2)   # a chain that terminates at its first trial is run again, one trial long
    assert run.status == _lib.ZF_CONVERGED and len(rows) == 1
    assert rows[0, _lib.TR_ERR] <= 1e-15
    Fstar = rows[0, _lib.TR_F]
    run.solver.close()

    K = 30
    o.update(tol=0.0, max_iter=K)
    run = NativeRun(prob, torch.zeros(n, dtype=torch.float64, device="cuda"), o)
    rows = run.advance(K)
    assert run.status == _lib.ZF_MAXITER and len(rows) == K
    assert np.all(rows[:, _lib.TR_TRIALS] == 1)
    F = rows[:, _lib.TR_F]
    assert np.all(F >= Fstar * (1 - 1e-12))
    assert F[-1] - Fstar <= 1e-6 * abs(Fstar)
    xK = run.solver.get_x()
    run.solver.close()
    idx = np.arange(0, n, 1000003)
    ds, cs = d[idx].cpu().numpy(), c[idx].cpu().numpy()
    ref = P.DiagQuadL1Ref(ds, cs, lam)
    # elementwise recursion on the sample (no reductions enter the iterate when no trial is rejected)
    from oracle import cpu_ref

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), np.zeros(idx.size), lr=0.45, nesterov=True,
                                                 tol=0.0, max_iter=K)
    assert np.array_equal(xK[idx], exp.x)


def test_verbose_prints_a_row_per_iteration(capsys):
    """verbose=True: the header of proximal_gradient.py:24-30 and one five-column row per outer
    iteration (the reference's formatter raises IndexError at :511-520 - documented deviation),
    on the device-resident path with chained passes, the one-trial path and the generic path."""
    from oracle import problems_ref as P
    from zfista_amd import minimize_proximal_gradient

    prob, ref = _problem(3001, seed=2)
    for cbs, extra in ((prob.callbacks(), {}), (prob.callbacks(), dict(return_all=True)),
                       (tuple((lambda fn: (lambda *a: fn(*a)))(fn) for fn in ref.callbacks()), {})):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = minimize_proximal_gradient(*cbs, np.zeros(3001), lr=0.45, nesterov=True, tol=0.0, max_iter=11,
                                             verbose=True, **extra)
        out = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("|")]
        assert "niter" in out[0] and "learning rate" in out[0] and set(out[1]) <= set("|-")
        rows = out[2:]
        assert len(rows) == res.nit == 11
        assert [int(r.split("|")[1]) for r in rows] == list(range(1, 12))
        assert all(len(r.split("|")) == 7 for r in rows)
