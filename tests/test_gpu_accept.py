"""GPU: acceptance="resolved" - the sufficient-decrease test of zfista/proximal_gradient.py:303 evaluated below ulp(F).

The reference accepts a trial when F(x+) - F(x_k) <= fun + tol_internal, fun = <grad f(y), x+ - y> + g(x+) +
|x+ - y|^2 / 2 / lr + (f(y) - F(x_k)) (:149-155): two differences of O(|F|) sums.  Once |x+ - y|^2 is far below
ulp(F) the outcome is rounding noise - the reference's own run shows it (fixture G12: rejections at iterations 90, 94,
95, 97, 102 at n = 1e7), and a long solve at n = 1e8 ends in "Backtracking failed".  The opt-in mode evaluates the SAME
inequality with F(x_k) and g(x+) cancelled and f(x+) - f(y) accumulated element by element (a second instantiation of
the chain kernels: zf_elem_diag<..., RES>; pack slot 7; zf_eval_trial).  Checked here:
  (i)   below the noise floor both modes take the same decisions: iterates bit-identical, traces to rounding - against
        the reference's own outputs (G3, G12 up to iteration 89) and the oracle's restatement of the resolved form;
  (ii)  across it, every decision of the resolved mode is the decision :303 takes in EXTENDED precision on the CPU, a
        1000-iteration solve at n = 1e7 ends in "maximum iterations", not in "Backtracking failed";
  (iii) chains of 16, single trials, run-ahead passes, passes ahead through a communicator and sharded solves agree bit
        for bit in that mode too; the result says which test ran."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_backtrack_iter=100, decay_rate=0.5,
            nesterov=False, nesterov_ratio=(0, 0.25), deprecated=False, return_all=False)


def _prob(n, seed, bounds=None):
    from oracle import problems_ref as P
    from zfista_amd.problems import DiagQuadL1

    d, c, lam = P.make_pdiag(n, seed=seed)
    return DiagQuadL1(d, c, lam, bounds=bounds), P.DiagQuadL1Ref(d, c, lam), (d, c, lam)


def _run(prob, x0, opts, sub=16, chunk=64, acceptance="resolved"):
    from zfista_amd import _lib
    from zfista_amd.proximal_gradient import NativeRun

    o = dict(BASE)
    o.update(opts)
    o["sub_iters"] = sub
    o["acceptance"] = acceptance
    run = NativeRun(prob, x0, o)
    rows = [np.zeros((0, _lib.ZF_TRACE_COLS))]
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(chunk))
    ctl = run.solver.ctl
    out = dict(rows=np.concatenate(rows), x=run.solver.get_x(), nit=int(ctl.nit), status=int(ctl.status), lr=ctl.lr, F=ctl.F_old,
               trials=int(ctl.total_trials), mode=int(ctl.accept_mode), report=run.solver.ahead_report())
    run.solver.close()
    return out


def _same(a, b):
    assert (a["nit"], a["status"], a["lr"], a["F"], a["trials"]) == (b["nit"], b["status"], b["lr"], b["F"], b["trials"])
    assert np.array_equal(a["rows"], b["rows"]) and np.array_equal(a["x"], b["x"])


@pytest.mark.parametrize("case", [
    dict(n=10007, lr=4.0, nesterov=True, tol=1e-9, max_iter=200),
    dict(n=10007, lr=4.0, nesterov=False, tol=1e-9, max_iter=400),
    dict(n=50001, lr=16.0, nesterov=True, nesterov_ratio=(0.5, 1 / 16), tol=0.0, max_iter=90),
    dict(n=4099, lr=4.0, nesterov=True, tol=0.0, max_iter=120, bounds=(-0.3, 0.4)),
    dict(n=30011, lr=4.0, nesterov=True, tol=0.0, max_iter=80, deprecated=True),
])
def test_below_the_noise_floor_both_tests_take_the_same_decisions(case):
    """Where the reference's evaluation resolves the test, the resolved evaluation decides the same: identical trial
    counts, step sizes and ITERATES bit for bit (the iterate arithmetic does not depend on the mode); the reported F
    values agree to rounding (f(x+) is formed as f(y) + [f(x+) - f(y)] in that mode).  Also against the oracle's
    restatement of the resolved form (cpu_ref f_diff=)."""
    from oracle import cpu_ref
    from zfista_amd import _lib

    c = dict(case)
    n, bounds = c.pop("n"), c.pop("bounds", None)
    prob, ref, _ = _prob(n, 5, bounds)
    x0 = np.random.default_rng(n).standard_normal(n)
    if bounds is not None:
        # (a start OUTSIDE the box has F(x_0) = inf, and the reference's expression then accepts any first trial -
        #  -inf <= -inf - where the resolved form tests the smooth part as always: the one place the two differ by design)
        x0 = np.clip(x0, *bounds)
    a = _run(prob, x0, c, acceptance="reference")
    b = _run(prob, x0, c, acceptance="resolved")
    assert (a["mode"], b["mode"]) == (_lib.ZF_ACCEPT_REFERENCE, _lib.ZF_ACCEPT_RESOLVED)
    assert (a["nit"], a["status"], a["lr"], a["trials"]) == (b["nit"], b["status"], b["lr"], b["trials"])
    assert np.array_equal(a["x"], b["x"])
    for col in (_lib.TR_ERR, _lib.TR_LR, _lib.TR_TRIALS, _lib.TR_GX, _lib.TR_FY):
        assert np.array_equal(a["rows"][:, col], b["rows"][:, col]), col
    np.testing.assert_allclose(b["rows"][:, _lib.TR_F], a["rows"][:, _lib.TR_F], rtol=1e-13)
    one = _run(prob, x0, c, sub=1)                 # single trials in the resolved mode: the same solve
    _same(one, b)
    if bounds is None:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kw = dict(BASE, **c)
            kw.pop("return_all")
            exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), x0, f_diff=ref.f_diff, **kw)
        assert exp.nit == b["nit"] and list(exp.alltrials) == list(b["rows"][:, _lib.TR_TRIALS].astype(int))
        assert np.array_equal(exp.x, b["x"])


def test_against_the_reference_run_up_to_its_noise_floor(golden):
    """G12: the imported reference on P-diag n = 1e7, lr 0.45, 110 FISTA iterations - its lr / trial sequence is clean up
    to iteration 89 and rounding noise from 90 on.  The resolved mode reproduces the clean part exactly (iterates 1e-10 as
    every parity test, here they are bit-identical to the default mode's) and stays clean where the reference stumbles."""
    from oracle import problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.problems import DiagQuadL1

    G = golden("g12_noise_floor_diag_n1e7.npz")
    n = 10_000_000
    d, c, lam = P.make_pdiag(n, seed=1)
    prob = DiagQuadL1(d, c, lam)
    opts = dict(lr=0.45, nesterov=True, tol=0.0, max_iter=110)
    res = _run(prob, np.zeros(n), opts)
    ref = _run(prob, np.zeros(n), opts, acceptance="reference")
    trials = res["rows"][:, _lib.TR_TRIALS].astype(int)
    assert int(G("first_rejection")) == 90 and list(trials[:89]) == [1] * 89
    np.testing.assert_allclose(res["rows"][:89, _lib.TR_F], np.asarray(G("allfuns"))[1:90], rtol=1e-10)
    np.testing.assert_allclose(res["rows"][:89, _lib.TR_ERR], np.asarray(G("allerrs"))[:89], rtol=1e-10)
    # across the floor: the reference's evaluation rejects (the reference itself at 90, 94, 95, 97, 102; this engine's
    # sums at their own places), the resolved one never does on this problem - lr 0.45 < 1 / max d satisfies the test
    assert list(trials) == [1] * 110 and res["lr"] == 0.45
    assert ref["trials"] > ref["nit"], "the default mode is expected to show the reference's noise-floor rejections"


def _extended_test(d, c, lam, x_old, y, x_new, lr, tol_internal):
    """zfista/proximal_gradient.py:303 evaluated in extended precision (x87 long double: 64-bit mantissa; pairwise sums):
    returns F(x+) - F(x_k) - fun - tol_internal, accepted iff <= 0."""
    L = np.longdouble
    d, c, x_old, y, x_new = (np.asarray(v, dtype=L) for v in (d, c, x_old, y, x_new))
    f = lambda x: L(0.5) * np.sum(d * ((x - c) * (x - c)))        # noqa: E731
    g = lambda x: L(lam) * np.sum(np.abs(x))                      # noqa: E731
    grad = d * (y - c)
    step = x_new - y
    fun = np.sum(grad * step) + g(x_new) + np.sum(step * step) / 2 / L(lr) + (f(y) - (f(x_old) + g(x_old)))
    return float((f(x_new) + g(x_new)) - (f(x_old) + g(x_old)) - fun - L(tol_internal))


def test_across_the_noise_floor_every_decision_is_the_one_extended_precision_takes():
    """n = 1e6, lr above 1 / L for the first line search, 400 FISTA iterations: far beyond where the default evaluation
    stops resolving the test.  Every accepted iteration of the resolved mode must satisfy :303 evaluated in extended
    precision on the CPU at the recorded iterates, and (sampled) the step size before each halving must violate it."""
    from oracle import cpu_ref
    from zfista_amd import _lib, minimize_proximal_gradient

    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("needs an extended-precision long double on the host")
    n = 1_000_003
    prob, ref, (d, c, lam) = _prob(n, 9)
    kw = dict(lr=1.7, nesterov=True, tol=0.0, max_iter=400, return_all=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*prob.callbacks(), np.zeros(n), acceptance="resolved", **kw)
        dflt = minimize_proximal_gradient(*prob.callbacks(), np.zeros(n), **kw)
    assert res["acceptance"] == "resolved" and "acceptance" not in dflt
    assert res.nit == 400
    betas = cpu_ref.momentum_sequence(401)
    lr_final = None
    worst = -np.inf
    checked = 0
    x_prev, y = np.zeros(n), np.zeros(n)
    # the step sizes: the run's trace is not part of the result; recompute them from the iterates - lr only ever
    # halves from 1.7, and the accepted lr of iteration k is the largest such value that reproduces x_k from y
    lr = 1.7
    for k in range(1, 401):
        x_new = np.asarray(res.allvecs[k])
        while True:   # which halving produced this iterate (bit-exact iterate arithmetic)
            v = y - lr * (d * (y - c))
            cand = np.sign(v) * np.maximum(np.abs(v) - lam * lr, 0.0)
            if np.array_equal(cand, x_new):
                break
            # a rejected step size: :303 must fail for it in extended precision (sampled - it is a full evaluation)
            if checked < 40:
                assert _extended_test(d, c, lam, x_prev, y, cand, lr, 1e-12) > 0, (k, lr)
            lr *= 0.5
            assert lr > 1e-6, f"iteration {k}: no step size reproduces the iterate"
        if k % 7 == 0 or k > 380:
            val = _extended_test(d, c, lam, x_prev, y, x_new, lr, 1e-12)
            worst = max(worst, val)
            assert val <= 0.0, f"iteration {k}: accepted although :303 fails in extended precision ({val:.3e})"
            checked += 1
        y = x_new + betas[k - 1] * (x_new - x_prev)   # beta_k: the factor applied after outer iteration k
        x_prev = x_new
        lr_final = lr
    assert checked >= 60 and lr_final < 1.7
    # the default evaluation meanwhile has lost steps to rounding noise (or not yet, at this size: then both agree)
    assert dflt.nit <= 400


def test_a_long_solve_ends_at_max_iter_not_in_backtracking_failed():
    """n = 1e7, 1000 iterations: with the reference's evaluation the step size collapses through rounding-noise
    rejections ('Backtracking failed' at iteration 361 at n = 1e8; at 1e7 lr is halved dozens of times);
    resolved, not a single trial of this problem is rejected (lr 0.45 satisfies the test for every x)."""
    from zfista_amd import _lib

    n = 10_000_000
    prob, _, _ = _prob(n, 1)
    opts = dict(lr=0.45, nesterov=True, tol=0.0, max_iter=1000)
    res = _run(prob, np.zeros(n), opts)
    assert res["status"] == _lib.ZF_MAXITER and res["nit"] == 1000 and res["trials"] == 1000 and res["lr"] == 0.45
    dflt = _run(prob, np.zeros(n), opts, acceptance="reference")
    assert dflt["trials"] > dflt["nit"] or dflt["status"] == _lib.ZF_BACKTRACK_FAILED
    assert res["F"] <= dflt["F"] * (1 + 1e-12)


@pytest.mark.parametrize("n,opts", [
    (300_001, dict(lr=16.0, nesterov=True, tol=0.0, max_iter=300)),
    (2_000_003, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=200)),
    (300_001, dict(lr=0.45, nesterov=True, tol=1e-7, max_iter=5000)),
])
def test_every_way_of_running_a_pass_agrees_in_the_resolved_mode(n, opts, monkeypatch):
    """Chains of 16 with run-ahead passes (the default on these grids), one launch per pass, single trials, and the solve
    through a 1-rank RCCL communicator with sharded run-ahead passes and with passes ahead: one result, bit for bit."""
    import torch

    from zfista_amd.comm import LibComm
    from zfista_amd.problems import DiagQuadL1

    prob, _, (d, c, lam) = _prob(n, 13)
    x0 = np.zeros(n)
    ra = _run(prob, x0, opts, chunk=8)
    assert ra["report"]["runahead_overlapped"] >= 1
    monkeypatch.setenv("ZF_RUNAHEAD", "0")
    plain = _run(prob, x0, opts, chunk=8)
    assert plain["report"]["runahead"] == 0
    _same(ra, plain)
    monkeypatch.delenv("ZF_RUNAHEAD")
    _same(_run(prob, x0, opts, sub=1), plain)
    for scheme in ("run-ahead", "ahead"):   # (behind a communicator a one-round grid takes sharded run-ahead passes; else passes ahead)
        if scheme == "ahead":
            monkeypatch.setenv("ZF_RUNAHEAD_SHARDED", "0")
        comm = LibComm(0, 1, LibComm.new_unique_id())
        sharded = _run(DiagQuadL1(d, c, lam, group=comm), x0, opts, chunk=8)
        torch.cuda.synchronize()
        comm.close()
        assert sharded["report"]["ahead" if scheme == "ahead" else "runahead"] >= 2, (scheme, sharded["report"])
        _same(sharded, plain)


def test_the_mode_is_refused_where_it_does_not_exist():
    from oracle import problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import LeastSquaresL1

    A, b, lam = P.make_plasso(32, 64, seed=0)
    with pytest.raises(ValueError, match="separable"):
        minimize_proximal_gradient(*LeastSquaresL1(A, b, lam).callbacks(), np.zeros(64), acceptance="resolved", max_iter=3)
    with pytest.raises(ValueError, match="acceptance"):
        minimize_proximal_gradient(*LeastSquaresL1(A, b, lam).callbacks(), np.zeros(64), acceptance="exact", max_iter=3)
