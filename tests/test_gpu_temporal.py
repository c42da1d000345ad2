"""GPU parity of temporal blocking (P-diag): one pass of the fused kernel computes S
consecutive iterations in registers and the decide step examines them in order, discarding
the speculative rest at the first rejection / termination (csrc/zf_kernels_step.h).  The
results must not depend on S: every trace row, branch decision and iterate is compared
bit for bit across S in {1, 2, 4, 8, 16} and against the golden vectors of the reference solver
(zfista/proximal_gradient.py:279-307,510,525-543)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_backtrack_iter=100, decay_rate=0.5,
            nesterov=False, nesterov_ratio=(0, 0.25), deprecated=False, return_all=False)

RUNS = {
    "fista_lr0.45": dict(lr=0.45, nesterov=True, tol=0.0, max_iter=60),
    "ista_lr0.45": dict(lr=0.45, nesterov=False, tol=0.0, max_iter=60),
    "fista_lr4_backtrack": dict(lr=4.0, nesterov=True, tol=0.0, max_iter=60),
    "fista_tol1e-6": dict(lr=0.45, nesterov=True, tol=1e-6, max_iter=10000),
    "fista_ab_0.5_0.25": dict(lr=0.45, nesterov=True, nesterov_ratio=(0.5, 0.25), tol=0.0, max_iter=60),
}


def _run(prob, x0, opts, sub, chunk=5):
    from zfista_amd import _lib
    from zfista_amd.proximal_gradient import NativeRun

    o = dict(BASE)
    o.update(opts)
    o["sub_iters"] = sub
    run = NativeRun(prob, x0, o)
    assert run.sub_iters == sub
    rows = [np.zeros((0, _lib.ZF_TRACE_COLS))]
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(chunk))
    x = run.solver.get_x()
    ctl = run.solver.ctl
    out = dict(rows=np.concatenate(rows), x=x, nit=int(ctl.nit), status=int(ctl.status), lr=ctl.lr,
               F=ctl.F_old, trials=int(ctl.total_trials))
    run.solver.close()
    return out


def _pdiag(n, seed=1, box=None):
    from oracle import problems_ref as P
    from zfista_amd.problems import DiagQuadL1

    d, c, lam = P.make_pdiag(n, seed=seed)
    if box is None:
        return DiagQuadL1(d, c, lam)
    return DiagQuadL1(d, c, lam, bounds=box)


@pytest.mark.parametrize("sub", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("tag", list(RUNS))
def test_temporal_blocking_matches_reference_golden(tag, sub, golden):
    from zfista_amd import _lib

    G = golden("g3_diag_n10007.npz")
    r = _run(_pdiag(10007), np.zeros(10007), RUNS[tag], sub)
    assert r["nit"] == int(G(f"{tag}.nit"))
    rows = r["rows"]
    assert len(rows) == r["nit"]
    np.testing.assert_allclose(rows[:, _lib.TR_ERR], G(f"{tag}.allerrs"), rtol=1e-10, atol=0)
    np.testing.assert_allclose(rows[:, _lib.TR_F], G(f"{tag}.allfuns")[1:], rtol=1e-10, atol=0)
    assert np.array_equal(rows[:, _lib.TR_LR], G(f"{tag}.alllrs"))
    assert np.array_equal(rows[:, _lib.TR_TRIALS].astype(np.int64), G(f"{tag}.alltrials"))
    assert np.array_equal(r["x"], G(f"{tag}.x")), "iterates are expected bit-identical to the reference"


CASES = [
    # (n, options): termination in the middle of a pass, rejections at every position of a pass,
    # max_iter not a multiple of S, single-element and ragged tails, box constraint
    (10007, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=37)),
    (10007, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=1)),
    (10007, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=2)),
    (10007, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=3)),
    (10007, dict(lr=64.0, nesterov=True, tol=0.0, max_iter=23)),
    (10007, dict(lr=64.0, nesterov=False, tol=1e-9, max_iter=500)),
    (10007, dict(lr=3.0, nesterov=True, decay_rate=0.9, tol=1e-8, max_iter=400)),
    (10007, dict(lr=0.45, nesterov=True, decay_rate=1.0, tol=1e-7, max_iter=300)),
    (10007, dict(lr=0.45, nesterov=True, deprecated=True, tol=1e-7, max_iter=300)),
    (10007, dict(lr=1e6, nesterov=True, max_backtrack_iter=3, tol=0.0, max_iter=50)),   # backtracking fails
    (1, dict(lr=0.45, nesterov=True, tol=1e-12, max_iter=200)),
    (3, dict(lr=0.45, nesterov=True, tol=1e-12, max_iter=200)),
    (2049, dict(lr=0.45, nesterov=True, tol=1e-12, max_iter=77)),
    (1 << 20, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=41)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_temporal_blocking_invariance(case):
    n, opts = CASES[case]
    prob = _pdiag(n, seed=3 + case)
    x0 = np.random.default_rng(case).standard_normal(n)
    ref = _run(prob, x0, opts, 1)
    assert ref["nit"] >= 1 or ref["status"] != 0
    for sub in (2, 4, 8, 16):
        for chunk in (1, 7):
            r = _run(prob, x0, opts, sub, chunk=chunk)
            assert (r["nit"], r["status"]) == (ref["nit"], ref["status"]), (sub, chunk)
            assert np.array_equal(r["rows"], ref["rows"]), (sub, chunk)
            assert np.array_equal(r["x"], ref["x"]), (sub, chunk)
            assert r["lr"] == ref["lr"] and r["F"] == ref["F"]
            assert r["trials"] == ref["trials"], "speculative sub-iterations must not count as trials"


@pytest.mark.parametrize("nesterov", [False, True])
def test_temporal_blocking_box(nesterov):
    n = 30011
    prob = _pdiag(n, seed=11, box=(-0.05, 0.07))
    x0 = np.zeros(n)
    opts = dict(lr=2.0, nesterov=nesterov, tol=1e-9, max_iter=150)
    ref = _run(prob, x0, opts, 1)
    for sub in (2, 4, 8, 16):
        r = _run(prob, x0, opts, sub)
        assert (r["nit"], r["status"]) == (ref["nit"], ref["status"])
        assert np.array_equal(r["rows"], ref["rows"])
        assert np.array_equal(r["x"], ref["x"])


def test_temporal_blocking_long_run_ring_wrap():
    """More iterations than ZF_RING: momentum / trace rings wrap under S = 4."""
    from zfista_amd import _lib

    n = 4099
    prob = _pdiag(n, seed=5)
    opts = dict(lr=0.45, nesterov=True, tol=0.0, max_iter=2 * _lib.ZF_RING + 13)
    ref = _run(prob, np.zeros(n), opts, 1, chunk=256)
    for sub in (4, 8, 16):
        r = _run(prob, np.zeros(n), opts, sub, chunk=256)
        assert r["nit"] == ref["nit"] == 2 * _lib.ZF_RING + 13
        assert np.array_equal(r["rows"], ref["rows"])
        assert np.array_equal(r["x"], ref["x"])


def test_set_max_iter_resumes_a_solve():
    from zfista_amd import _lib
    from zfista_amd.proximal_gradient import NativeRun

    n = 10007
    prob = _pdiag(n)
    ref = _run(prob, np.zeros(n), dict(lr=0.45, nesterov=True, tol=0.0, max_iter=30), 4)
    o = dict(BASE)
    o.update(lr=0.45, nesterov=True, tol=0.0, max_iter=10, sub_iters=4)
    run = NativeRun(prob, np.zeros(n), o)
    rows = []
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(3))
    assert run.status == _lib.ZF_MAXITER and run.nit_seen == 10
    run.set_max_iter(30)
    assert run.status == _lib.ZF_RUNNING
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(3))
    assert run.status == _lib.ZF_MAXITER and run.nit_seen == 30
    assert np.array_equal(np.concatenate(rows), ref["rows"])
    assert np.array_equal(run.solver.get_x(), ref["x"])
    run.solver.close()


def test_large_n_geometry_is_a_function_of_n_only():
    """Above 4096 tiles the kernel runs several tiles per workgroup.  That geometry fixes the
    rounding of the reduced sums, so it must not depend on the chain length (or on a timing
    measurement, as it did in round 1): S = 1 and S = 8 give identical trace rows and iterates at
    n = 9e6 too, and a second solve of the same problem repeats the first bit for bit."""
    n = 9_000_001
    prob = _pdiag(n, seed=4)
    x0 = np.zeros(n)
    opts = dict(lr=3.0, nesterov=True, tol=0.0, max_iter=29)   # backtracks first, then chains
    ref = _run(prob, x0, opts, 1, chunk=64)
    again = _run(prob, x0, opts, 8, chunk=64)
    third = _run(prob, x0, opts, 8, chunk=3)
    fourth = _run(prob, x0, opts, 16, chunk=5)
    for r in (again, third, fourth):
        assert (r["nit"], r["status"], r["lr"], r["F"], r["trials"]) == \
            (ref["nit"], ref["status"], ref["lr"], ref["F"], ref["trials"])
        assert np.array_equal(r["rows"], ref["rows"])
        assert np.array_equal(r["x"], ref["x"])


@pytest.mark.parametrize("sub", [1, 8])
@pytest.mark.parametrize("slots", [None, 19])
def test_streaming_return_all_records_every_iterate(sub, slots):
    """return_all on the device-resident path: every trial stores its iterate into a ring in HBM as
    it computes it (zf_solver_set_history), chains stay 8 long, the host receives the trace rows per
    chunk and the iterates on access.  A run with rejections (speculative iterates of a broken chain
    are overwritten by the retry), terminated by tol in the middle of a chain; with a ring smaller
    than the run (19 slots: iterates are moved to the host in blocks) and a roomy one - all against
    the oracle's allvecs, bit for bit."""
    import warnings

    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.proximal_gradient import NativeRun

    n = 10007
    d, c, lam = P.make_pdiag(n, seed=1)
    kw = dict(lr=4.0, nesterov=True, tol=1e-6, max_iter=400)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n), return_all=True, **kw)
    o = dict(BASE)
    o.update(kw, return_all=True, sub_iters=sub, history_slots=slots)
    x0 = np.zeros(n)
    run = NativeRun(_pdiag(n), x0, o)
    assert run.sub_iters == sub
    rows = []
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(3))
    H = run.history()
    assert len(H) == exp.nit + 1 == len(exp.allvecs) and H[0] is x0
    for k in range(len(H)):
        assert np.array_equal(H[k], exp.allvecs[k]), k
    assert np.array_equal(H[-1], run.solver.get_x())
    np.testing.assert_allclose(np.concatenate(rows)[:, _lib.TR_ERR], exp.allerrs, rtol=1e-10)
    if slots:
        assert len(run._hist_host) >= exp.nit - slots   # the ring wrapped: older iterates live on the host
    run.solver.close()


def test_streaming_return_all_through_the_public_entry(golden):
    """minimize_proximal_gradient(..., return_all=True): allvecs is a sequence of NumPy arrays like
    the reference's list; the chains are 8 long (no per-iteration host round trip)."""
    import warnings

    from zfista_amd import minimize_proximal_gradient

    G = golden("g3_diag_n10007.npz")
    tag = "fista_lr4_backtrack"
    p = _pdiag(10007)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*p.callbacks(), np.zeros(10007), return_all=True, **RUNS[tag])
    assert res.nit == int(G(f"{tag}.nit")) and len(res.allvecs) == res.nit + 1
    assert np.array_equal(np.stack([res.allvecs[k] for k in G(f"{tag}.kept")]), G(f"{tag}.vecs"))
    assert np.array_equal(np.asarray(res.allvecs)[-1], res.x)
    np.testing.assert_allclose(res.allfuns, G(f"{tag}.allfuns"), rtol=1e-10)
