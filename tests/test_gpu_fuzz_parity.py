"""GPU: randomised option / size sweep of the two device-resident operators against the CPU
oracle (NumPy restatement of zfista/proximal_gradient.py, pinned to the imported reference by
tests/test_oracle_golden.py).  Every case compares the full result: nit, success / status /
message, per-iteration lr and trial counts, F and err traces, and the final iterate to 1e-10 -
with chained passes (S = 8, the default) AND one iteration per pass (return_all)."""
import warnings

import numpy as np
import pytest

from conftest import rel_err

import os

pytestmark = pytest.mark.gpu

TOL = 1e-10
# ZF_FUZZ_SCALE=10 runs ten times as many random cases (one-off soak; the default keeps the suite short)
SCALE = int(os.environ.get("ZF_FUZZ_SCALE", "1"))


def _options(rng):
    o = dict(
        lr=float(10 ** rng.uniform(-2, 1.5)),
        tol=float(rng.choice([0.0, 1e-4, 1e-7, 1e-10])),
        max_iter=int(rng.integers(1, 90)),
        max_backtrack_iter=int(rng.choice([1, 2, 5, 100])),
        decay_rate=float(rng.choice([0.3, 0.5, 0.9, 1.0])),
        nesterov=bool(rng.integers(0, 2)),
        nesterov_ratio=tuple(map(float, rng.choice([(0, 0.25), (0.5, 0.25), (0.25, 1 / 64)]))),
        deprecated=bool(rng.random() < 0.2),
        tol_internal=float(rng.choice([1e-12, 1e-6, 0.0])),
    )
    return o


def _compare(prob, ref, x0, o):
    from oracle import cpu_ref
    from zfista_amd import _lib, minimize_proximal_gradient
    from zfista_amd.proximal_gradient import NativeRun

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, return_all=True, **o)
        # Once F stops changing at double resolution the acceptance test (:303) compares rounding
        # residues: its outcome then depends on the summation order, for the reference (NumPy's
        # pairwise sums) as for any other implementation.  The sweep stays below that
        # point: the run is cut one iteration before F first stagnates.
        F = np.asarray(exp.allfuns, dtype=np.float64)
        stalled = np.flatnonzero(np.abs(np.diff(F)) <= 64 * np.finfo(float).eps * np.maximum(1.0, np.abs(F[1:])))
        if stalled.size and stalled[0] + 1 <= exp.nit:
            if stalled[0] < 1:
                pytest.skip("x0 is already at the resolution limit of the acceptance test")
            o = dict(o, max_iter=int(stalled[0]))
            exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, return_all=True, **o)
        res = minimize_proximal_gradient(*prob.callbacks(), x0, return_all=True, **o)       # S = 1
        res8 = minimize_proximal_gradient(*prob.callbacks(), x0, return_all=False, **o)     # chains
    for r in (res, res8):
        assert r.success == exp.success and r.message == exp.message, (r.message, exp.message)
        assert r.nit == exp.nit
        assert ("status" in r) == ("status" in exp) and r.get("status") == exp.get("status")
        scale = max(np.linalg.norm(exp.x), 1e-300)
        assert np.linalg.norm(r.x - exp.x) <= TOL * scale or np.array_equal(r.x, exp.x)
        np.testing.assert_allclose(r.fun, exp.fun, rtol=TOL, atol=0)
    assert np.array_equal(res8.x, res.x), "chained passes must not change the iterate"
    if exp.allerrs is not None and len(exp.allerrs):
        # err = max|x+ - y| is a difference of iterates that agree to 1e-10 of their size
        np.testing.assert_allclose(res.allerrs, exp.allerrs, rtol=1e-9,
                                   atol=TOL * max(1e-300, float(np.max(np.abs(np.stack(exp.allvecs))))))
        np.testing.assert_allclose(res.allfuns, exp.allfuns, rtol=TOL)
    # lr / trial-count sequence of the chained run (trace ring) against the oracle's
    full = dict(max_iter_internal=100000, warm_start=False, verbose=False, return_all=False) | o
    run = NativeRun(prob, x0, full)
    rows = [np.zeros((0, _lib.ZF_TRACE_COLS))]
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(3))
    rows = np.concatenate(rows)
    k = len(rows)
    assert k == len(exp.alllrs[:k]) and (k == exp.nit or not exp.success)
    assert np.array_equal(rows[:, _lib.TR_LR], np.asarray(exp.alllrs[:k], float))
    assert np.array_equal(rows[:, _lib.TR_TRIALS].astype(np.int64), np.asarray(exp.alltrials[:k], np.int64))
    run.solver.close()


@pytest.mark.parametrize("seed", range(40 * SCALE))
def test_fuzz_diag_quad_l1(seed):
    from oracle import problems_ref as P
    from zfista_amd.problems import DiagQuadL1

    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 7, 64, 255, 1024, 2049, 5000, 20011]))
    d, c, lam = P.make_pdiag(n, seed=seed)
    lam = float(rng.choice([0.0, 0.1, 1.5]))
    o = _options(rng)
    x0 = rng.standard_normal(n) * rng.choice([0.0, 1.0, 100.0])
    _compare(DiagQuadL1(d, c, lam), P.DiagQuadL1Ref(d, c, lam), x0, o)


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_fuzz_least_squares_l1(seed):
    from oracle import problems_ref as P
    from zfista_amd.problems import LeastSquaresL1

    rng = np.random.default_rng(2000 + seed)
    m, n = [(3, 1), (8, 5), (16, 33), (40, 64), (64, 128), (33, 257)][seed % 6]
    A, b, lam = P.make_plasso(m, n, seed=seed, n_informative=min(n, 5))
    o = _options(rng)
    o["lr"] = float(10 ** rng.uniform(-4, 0))
    x0 = rng.standard_normal(n) * rng.choice([0.0, 1.0])
    _compare(LeastSquaresL1(A, b, lam), P.LeastSquaresL1Ref(A, b, lam), x0, o)


def test_nan_input_propagates_like_the_reference(capsys):
    """A NaN in the data poisons every sum: no trial is ever accepted (NaN <= x is False), the line
    search fails and the error-shaped result is returned (proximal_gradient.py:303-307, :493-509)."""
    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import DiagQuadL1

    n = 1000
    d, c, lam = P.make_pdiag(n, seed=4)
    c[17] = np.nan
    kw = dict(lr=0.45, nesterov=True, max_iter=10, max_backtrack_iter=6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n), **kw)
        res = minimize_proximal_gradient(*DiagQuadL1(d, c, lam).callbacks(), np.zeros(n), **kw)
    capsys.readouterr()
    assert res.success == exp.success is False and res.message == exp.message
    assert np.array_equal(res.x, exp.x)


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_fuzz_diag_quad_l1_resolved_acceptance(seed):
    """The same random sweep with acceptance="resolved" against the oracle's restatement of that form
    (oracle.cpu_ref.minimize_proximal_gradient(..., f_diff=): the sufficient-decrease test with F(x_k) and g(x+)
    cancelled and f(x+) - f(y) formed element by element).  The resolved form does not stall where the reference's
    does, so the runs are NOT cut at the stagnation point: nit, status, message, lr and trial sequences, chained and
    single-trial solves, iterates bit for bit (the iterate arithmetic is NumPy's in both)."""
    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import _lib, minimize_proximal_gradient
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([1, 2, 3, 7, 64, 255, 1024, 2049, 5000, 20011]))
    d, c, lam = P.make_pdiag(n, seed=seed)
    lam = float(rng.choice([0.0, 0.1, 1.5]))
    o = _options(rng)
    x0 = rng.standard_normal(n) * rng.choice([0.0, 1.0, 100.0])
    prob, ref = DiagQuadL1(d, c, lam), P.DiagQuadL1Ref(d, c, lam)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, return_all=True, f_diff=ref.f_diff, **o)
        res = minimize_proximal_gradient(*prob.callbacks(), x0, return_all=True, acceptance="resolved", **o)     # single trials
        res16 = minimize_proximal_gradient(*prob.callbacks(), x0, acceptance="resolved", **o)                    # chains of 16
    for r in (res, res16):
        assert r.success == exp.success and r.message == exp.message, (r.message, exp.message)
        assert r.nit == exp.nit and r.get("status") == exp.get("status") and r["acceptance"] == "resolved"
        # (a knife-edge decision may still fall differently: the two sum f(x+) - f(y) in different orders - but the sums are
        #  of the size of the step, and the sweep has not met one)
        assert np.array_equal(r.x, exp.x)
        np.testing.assert_allclose(r.fun, exp.fun, rtol=TOL, atol=0)
    full = dict(max_iter_internal=100000, warm_start=False, verbose=False, return_all=False, acceptance="resolved") | o
    run = NativeRun(prob, x0, full)
    rows = [np.zeros((0, _lib.ZF_TRACE_COLS))]
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(3))
    rows = np.concatenate(rows)
    k = len(rows)
    assert k == len(exp.alllrs[:k]) and (k == exp.nit or not exp.success)
    assert np.array_equal(rows[:, _lib.TR_LR], np.asarray(exp.alllrs[:k], float))
    assert np.array_equal(rows[:, _lib.TR_TRIALS].astype(np.int64), np.asarray(exp.alltrials[:k], np.int64))
    run.solver.close()
