"""GPU: checkpoint / resume of a device-resident solve.  The state is
(x_k, x_{k-1}, control block); a solve resumed from it - in a new solver object, through a file,
with the same or another chain length - continues bit for bit like the uninterrupted one, and
the resumed solve equals the ORACLE's uninterrupted solve (iterates 1e-10, lr / trial sequence exactly)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_iter_internal=100000, max_backtrack_iter=100,
            warm_start=False, decay_rate=0.5, nesterov=False, nesterov_ratio=(0, 0.25), return_all=False, verbose=False,
            deprecated=False)


def _drain(run, chunk=3):
    from zfista_amd import _lib

    rows = [np.zeros((0, _lib.ZF_TRACE_COLS))]
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(chunk))
    return np.concatenate(rows)


@pytest.mark.parametrize("kw", [
    dict(lr=0.45, nesterov=True, tol=0.0, max_iter=61),
    dict(lr=64.0, nesterov=True, tol=0.0, max_iter=45),          # snapshots land inside the backtracking phase
    dict(lr=3.0, nesterov=True, nesterov_ratio=(0.5, 0.25), decay_rate=0.9, tol=1e-9, max_iter=300),
    dict(lr=0.45, nesterov=False, tol=1e-8, max_iter=200),
])
@pytest.mark.parametrize("stop_after", [1, 2, 5])
def test_resume_diag_is_bit_identical(kw, stop_after, tmp_path):
    from oracle import problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    n = 20011
    d, c, lam = P.make_pdiag(n, seed=8)
    prob = DiagQuadL1(d, c, lam)
    x0 = np.random.default_rng(1).standard_normal(n)
    o = BASE | kw
    ref_run = NativeRun(prob, x0, o)
    ref_rows = _drain(ref_run)
    ref_x, ref_ctl = ref_run.solver.get_x(), ref_run.solver.ctl
    ref = (int(ref_ctl.nit), int(ref_ctl.status), ref_ctl.lr, int(ref_ctl.total_trials))
    ref_run.solver.close()

    first = NativeRun(prob, x0, o)
    head = [first.advance(1) for _ in range(stop_after)]          # passes, not iterations
    state = first.snapshot()
    first.solver.close()
    np.savez(tmp_path / "ckpt.npz", **state)                       # through a file
    state = dict(np.load(tmp_path / "ckpt.npz"))
    # the ORACLE's uninterrupted solve of the same problem (zfista/proximal_gradient.py:463-554 restated in NumPy):
    # what a resumed solve has to reproduce is the reference's run, not only this engine's own
    import warnings

    from conftest import rel_err
    from oracle import cpu_ref

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), x0,
                                                 **(o | dict(return_all=True)))
    for sub in (0, 1, 4):                                          # same chain length, or another
        run = NativeRun.from_snapshot(prob, state, o | dict(sub_iters=sub))
        rows = np.concatenate(head + [_drain(run)])
        ctl = run.solver.ctl
        assert (int(ctl.nit), int(ctl.status), ctl.lr, int(ctl.total_trials)) == ref, sub
        assert np.array_equal(rows, ref_rows), sub
        x = run.solver.get_x()
        assert np.array_equal(x, ref_x), sub
        # ... and against the oracle: iteration count, final status, the lr / trial sequence across the
        # snapshot, the error and objective traces, the final iterate
        assert int(ctl.nit) == exp.nit and (int(ctl.status) == _lib.ZF_CONVERGED) == bool(exp.success), sub
        assert np.array_equal(rows[:, _lib.TR_LR], np.asarray(exp.alllrs)), sub
        assert np.array_equal(rows[:, _lib.TR_TRIALS], np.asarray(exp.alltrials, float)), sub
        np.testing.assert_allclose(rows[:, _lib.TR_ERR], exp.allerrs, rtol=1e-10, atol=0)
        np.testing.assert_allclose(rows[:, _lib.TR_F], exp.allfuns[1:], rtol=1e-10, atol=0)
        assert rel_err(x, exp.x) <= 1e-10, sub
        run.solver.close()


def test_resume_with_a_larger_max_iter_continues_a_finished_solve():
    from oracle import problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    n = 5003
    d, c, lam = P.make_pdiag(n, seed=2)
    prob = DiagQuadL1(d, c, lam)
    o = BASE | dict(lr=0.45, nesterov=True, tol=0.0)
    full = NativeRun(prob, np.zeros(n), o | dict(max_iter=50))
    rows_full = _drain(full)
    part = NativeRun(prob, np.zeros(n), o | dict(max_iter=20))
    rows_a = _drain(part)
    assert part.status == _lib.ZF_MAXITER
    cont = NativeRun.from_snapshot(prob, part.snapshot(), o | dict(max_iter=50))
    assert cont.status == _lib.ZF_RUNNING and cont.nit_seen == 20
    rows_b = _drain(cont)
    assert np.array_equal(np.concatenate([rows_a, rows_b]), rows_full)
    assert np.array_equal(cont.solver.get_x(), full.solver.get_x())


@pytest.mark.parametrize("nesterov", [False, True])
def test_resume_least_squares_is_bit_identical(nesterov):
    from oracle import problems_ref as P
    from zfista_amd.problems import LeastSquaresL1
    from zfista_amd.proximal_gradient import NativeRun

    A, b, lam = P.make_plasso(96, 200, seed=4)
    prob = LeastSquaresL1(A, b, lam)
    o = BASE | dict(lr=1.0, nesterov=nesterov, tol=0.0, max_iter=40)
    ref_run = NativeRun(prob, np.zeros(200), o)
    ref_rows, ref_x = _drain(ref_run), ref_run.solver.get_x()
    first = NativeRun(prob, np.zeros(200), o)
    head = first.advance(17)
    run = NativeRun.from_snapshot(prob, first.snapshot(), o)
    rows = np.concatenate([head, _drain(run)])
    assert np.array_equal(rows, ref_rows)
    assert np.array_equal(run.solver.get_x(), ref_x)
    # the resumed solve against the oracle's uninterrupted one
    import warnings

    from conftest import rel_err
    from oracle import cpu_ref
    from zfista_amd import _lib

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.LeastSquaresL1Ref(A, b, lam).callbacks(), np.zeros(200),
                                                 **(o | dict(return_all=True)))
    assert len(rows) == exp.nit == 40
    assert np.array_equal(rows[:, _lib.TR_LR], np.asarray(exp.alllrs))
    np.testing.assert_allclose(rows[:, _lib.TR_ERR], exp.allerrs, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(rows[:, _lib.TR_F], exp.allfuns[1:], rtol=1e-10, atol=0)
    assert rel_err(run.solver.get_x(), exp.x) <= 1e-10


def test_a_snapshot_carries_its_launch_geometry(monkeypatch):
    """The tiles per workgroup decide the order of the reduced sums (knife-edge accept / reject decisions).  A snapshot says
    which geometry it was taken with; a resume under another one (another build of the library; here: ZF_TILES_PER_WG) warns
    and still continues with the same iterates."""
    import warnings

    from oracle import problems_ref as P
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    n = 3_000_001
    d, c, lam = P.make_pdiag(n, seed=5)
    prob = DiagQuadL1(d, c, lam)
    o = BASE | dict(lr=0.45, nesterov=True, tol=0.0, max_iter=64)
    first = NativeRun(prob, np.zeros(n), o)
    first.advance(2)
    state = first.snapshot()
    assert int(state["tiles_per_wg"]) == first.solver.tiles_per_wg >= 2
    first.solver.close()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        same = NativeRun.from_snapshot(prob, state, o)          # the same rule: no warning
    want = (_drain(same), same.solver.get_x())[1]
    monkeypatch.setenv("ZF_TILES_PER_WG", "1")
    with pytest.warns(UserWarning, match="tiles per workgroup"):
        other = NativeRun.from_snapshot(prob, state, o)
    _drain(other)
    assert np.array_equal(other.solver.get_x(), want)          # (clean regime: the iterates are elementwise arithmetic)
