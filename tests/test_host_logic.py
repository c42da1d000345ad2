"""CPU: the product's control logic without a GPU.

* zf_decide_host (the function the decide kernel runs) against the oracle's line
  search on identical reduced scalars;
* the host driver (NativeRun / _solve_native: chunking, momentum table, trace ring,
  result assembly, warnings, error results) driven by tests/fake_engine.FakeSolver,
  compared with the oracle and the golden vectors of the reference."""
import ctypes as C
import json
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN
from fake_engine import FakeProblem, FakeSolver
from oracle import cpu_ref, problems_ref as P
from zfista_amd import _lib
from zfista_amd.proximal_gradient import NativeRun, _solve_native

BASE = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_iter_internal=100000,
            max_backtrack_iter=100, warm_start=False, decay_rate=0.5, nesterov=False,
            nesterov_ratio=(0, 0.25), return_all=False, verbose=False, deprecated=False)


def _ctl(**kw):
    c = _lib.Control()
    c.lr, c.tol, c.tol_internal, c.decay_rate = 1.0, 1e-5, 1e-12, 0.5
    c.max_iter, c.max_backtrack, c.world, c.F_old = 100, 3, 1, 10.0
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _decide(c, pack, world=1):
    lib = _lib.load()
    trace = np.zeros((_lib.ZF_RING, _lib.ZF_TRACE_COLS))
    p = np.ascontiguousarray(np.asarray(pack, float))
    _lib.check(lib.zf_decide_host(C.byref(c), C.sizeof(c), C.c_void_p(_lib.ptr(p)), C.c_void_p(_lib.ptr(trace))))
    return trace


def test_decide_accept_reject_and_failure():
    # pack: f_y, dot, ss, g_x, f_x, err
    c = _ctl()
    # F_x = 9, fun = dot+g + ss/2/lr + f_y - F_old = -2+1+0.5+10.2-10 = -0.3 ; 9-10 <= -0.3 -> accept
    tr = _decide(c, [10.2, -2.0, 1.0, 1.0, 8.0, 0.5, 0, 0])
    assert (c.nit, c.trial, c.cur, c.status) == (1, 0, 1, _lib.ZF_RUNNING)
    assert c.F_old == 9.0 and c.lr == 1.0 and c.need_grad == 1
    assert tr[0, _lib.TR_ERR] == 0.5 and tr[0, _lib.TR_F] == 9.0 and tr[0, _lib.TR_TRIALS] == 1
    assert tr[0, _lib.TR_FUN] == (-2.0 + 1.0) + np.sqrt(1.0) ** 2 / 2 / 1.0 + (10.2 - 10.0)   # :150-155 order
    # reject: F_x - F_old = 0.5 > fun
    c = _ctl()
    _decide(c, [10.0, -2.0, 1.0, 1.0, 9.5, 0.5, 0, 0])
    assert (c.nit, c.trial, c.cur, c.lr, c.need_grad) == (0, 1, 0, 0.5, 0)
    _decide(c, [10.0, -2.0, 1.0, 1.0, 20.0, 0.5, 0, 0])
    _decide(c, [10.0, -2.0, 1.0, 1.0, 20.0, 0.5, 0, 0])
    assert c.status == _lib.ZF_BACKTRACK_FAILED and c.lr == 0.125 and c.nit == 0
    # further steps are no-ops
    _decide(c, [10.2, -2.0, 1.0, 1.0, 8.0, 0.5, 0, 0])
    assert c.nit == 0 and c.total_trials == 3


def test_decide_termination_order_and_flags():
    c = _ctl(max_iter=1)
    _decide(c, [10.2, -2.0, 1.0, 1.0, 8.0, 1e-6, 0, 0])      # err < tol wins over max_iter (:525 before :539)
    assert c.status == _lib.ZF_CONVERGED
    c = _ctl(max_iter=1)
    _decide(c, [10.2, -2.0, 1.0, 1.0, 8.0, 1e-5, 0, 0])      # strict <
    assert c.status == _lib.ZF_MAXITER
    c = _ctl(decay_rate=1.0)
    _decide(c, [0.0, 0.0, 0.0, 0.0, 99.0, 1.0, 0, 0])        # decay_rate == 1 accepts unconditionally (:298)
    assert c.nit == 1
    c = _ctl(deprecated=1)
    # deprecated: f_x - f_y <= dot + g + ss/2/lr + tol  (:301); 8-10.2=-2.2 <= -0.5
    _decide(c, [10.2, -2.0, 1.0, 1.0, 8.0, 1.0, 0, 0])
    assert c.nit == 1 and abs(c.fun - (-0.5)) < 1e-15
    c = _ctl(F_old=np.inf)                                     # x0 outside the box: everything is accepted
    _decide(c, [1.0, 0.0, 0.0, 0.0, 5.0, 1.0, 0, 0])
    assert c.nit == 1
    c = _ctl()
    _decide(c, [np.nan, 0.0, 0.0, 0.0, 5.0, 1.0, 0, 0])       # NaN never satisfies <= (np.all(nan<=..) is False)
    assert c.nit == 0 and c.trial == 1


def test_six_buffer_ring_never_hands_a_pass_its_predecessors_inputs():
    """ring_size 6 (run-ahead passes): a pass writes the two buffers BEHIND x_k in ring order, so the pass launched behind
    it - before it is decided - never overwrites what it reads, whatever mixture of whole chains, single trials and
    rejections came before; rings of 3 and 4 keep the lowest-numbered free buffers."""
    ok = [10.2, -2.0, 1.0, 1.0, 8.0, 0.5, 0, 0]       # accepted against F_old = 10 (test above)
    bad = [10.0, -2.0, 1.0, 1.0, 20.0, 0.5, 0, 0]     # rejected
    rng = np.random.default_rng(3)
    for sub in (1, 2, 4):
        c = _ctl(ring_size=6, sub_iters=sub, cur=0, prev=5, max_iter=10 ** 6, max_backtrack=50)
        prev_inputs = None
        for step in range(200):
            inputs = (c.cur, c.prev)
            accept_all = rng.random() < 0.7
            packs = []
            for j in range(sub):
                packs += ok if (accept_all or j < sub - 1) else bad
            lag0, nit0 = c.lag, c.nit
            c.F_old = 10.0
            _decide(c, packs)
            if (c.cur, c.prev) != inputs:   # the pass stored iterates: where?
                outputs = {c.cur, c.prev} - set(inputs) if sub == 1 else {c.cur, c.prev}
                assert not (outputs & set(inputs)), (sub, step, inputs, c.cur, c.prev)
                if prev_inputs is not None:
                    assert not (outputs & set(prev_inputs)), (sub, step, prev_inputs, inputs, c.cur, c.prev)
                assert c.cur == (inputs[0] + (1 if c.nit - nit0 + lag0 == 1 else 2)) % 6   # ring order
                prev_inputs = inputs
            c.lr = 1.0
    for ring, want in ((3, (1, 0)), (4, (2, 1))):
        c = _ctl(ring_size=ring, sub_iters=1 if ring == 3 else 2, cur=0, prev=ring - 1)
        _decide(c, ok * (1 if ring == 3 else 2))
        assert (c.cur, c.prev) == want


def test_decide_sums_packs_in_rank_order():
    c = _ctl(world=3)
    packs = np.array([[1.0, -1.0, 0.25, 0.5, 2.0, 0.1, 0, 0],
                      [2.0, -0.5, 0.25, 0.25, 3.0, 0.7, 0, 0],
                      [7.2, -0.5, 0.5, 0.25, 3.0, 0.3, 0, 0]])
    tr = _decide(c, packs.ravel())
    assert c.nit == 1 and tr[0, _lib.TR_ERR] == 0.7 and tr[0, _lib.TR_F] == 9.0
    assert tr[0, _lib.TR_FY] == (1.0 + 2.0) + 7.2


def _factory(fields, options, problem, x0):
    return FakeSolver(fields, options, problem, x0)


@pytest.mark.parametrize("tag,kw", [
    ("fista_lr0.45", dict(lr=0.45, nesterov=True, tol=0.0, max_iter=60)),
    ("ista_lr0.45", dict(lr=0.45, nesterov=False, tol=0.0, max_iter=60)),
    ("fista_lr4_backtrack", dict(lr=4.0, nesterov=True, tol=0.0, max_iter=60)),
    ("fista_tol1e-6", dict(lr=0.45, nesterov=True, tol=1e-6, max_iter=10000)),
    ("fista_ab_0.5_0.25", dict(lr=0.45, nesterov=True, nesterov_ratio=(0.5, 0.25), tol=0.0, max_iter=60)),
])
@pytest.mark.parametrize("return_all", [True, False])
def test_host_driver_against_golden(golden, tag, kw, return_all):
    G = golden("g3_diag_n10007.npz")
    d, c, lam = P.make_pdiag(10007, seed=1)
    prob = FakeProblem(d, c, lam)
    o = BASE | kw | dict(return_all=return_all)
    res, status = _solve_native(prob, np.zeros(10007), o, solver_factory=_factory)
    assert res.nit == int(G(f"{tag}.nit"))
    assert np.array_equal(res.x, G(f"{tag}.x"))
    np.testing.assert_allclose(res.fun, G(f"{tag}.fun"), rtol=1e-12)
    assert res.status == int(G(f"{tag}.status"))
    if return_all:
        assert np.array_equal(np.stack([res.allvecs[k] for k in G(f"{tag}.kept")]), G(f"{tag}.vecs"))
        np.testing.assert_allclose(res.allerrs, G(f"{tag}.allerrs"), rtol=0, atol=0)
        np.testing.assert_allclose(res.allfuns, G(f"{tag}.allfuns"), rtol=1e-12)
    else:
        assert res.allvecs is None and res.allfuns is None and res.allerrs is None


def test_host_driver_chunks_cross_the_ring():
    """More iterations than ZF_RING: momentum ring and trace ring wrap correctly."""
    n = 257
    d, c, lam = P.make_pdiag(n, seed=5)
    prob = FakeProblem(d, c, lam)
    K = 2 * _lib.ZF_RING + 77
    o = BASE | dict(lr=0.05, nesterov=True, tol=0.0, max_iter=K)
    run = NativeRun(prob, np.zeros(n), o, solver_factory=_factory)
    errs = []
    while run.status == _lib.ZF_RUNNING:
        errs.extend(run.advance(700)[:, _lib.TR_ERR])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n),
                                                 lr=0.05, nesterov=True, tol=0.0, max_iter=K, return_all=True)
    assert run.nit_seen == K and len(errs) == K
    assert np.array_equal(np.asarray(errs), np.asarray(exp.allerrs))
    assert np.array_equal(run.solver.get_x(), exp.x)


def test_host_driver_result_shapes_and_messages(capsys):
    """Result keys / messages / warnings equal the reference's (fixture G6)."""
    shapes = json.load(open(os.path.join(GOLDEN, "g6_result_shapes.json")))
    d, c, lam = P.make_pdiag(33, seed=2)
    prob = FakeProblem(d, c, lam)
    x0 = np.zeros(33)
    res, st = _solve_native(prob, x0, BASE | dict(lr=0.4, tol=1e-9), solver_factory=_factory)
    assert sorted(res.keys()) == shapes["success"]["keys"]
    assert (res.status, res.message, res.success) == (1, shapes["success"]["message"], True)
    res, st = _solve_native(prob, x0, BASE | dict(lr=0.4, max_iter=3, tol=0.0), solver_factory=_factory)
    assert sorted(res.keys()) == shapes["max_iter"]["keys"]
    assert (res.status, res.message, res.success, res.nit) == (0, shapes["max_iter"]["message"], False, 3)
    assert st == _lib.ZF_MAXITER
    # backtracking failure -> error-shaped result, printed not raised (:493-509)
    res, st = _solve_native(prob, x0, BASE | dict(lr=1e6, max_backtrack_iter=2), solver_factory=_factory)
    assert sorted(res.keys()) == shapes["backtracking_failure"]["keys"]
    assert res.message == shapes["backtracking_failure"]["message"] and res.nit == 0 and not res.success
    assert np.array_equal(res.x, x0)
    assert capsys.readouterr().out.splitlines()[-1] == shapes["backtracking_failure"]["stdout_ref"][0]
    res, st = _solve_native(prob, x0, BASE | dict(lr=0.4, max_backtrack_iter=0), solver_factory=_factory)
    assert res.message == shapes["backtracking_failure"]["message"]


def test_public_entry_warns_like_the_reference(monkeypatch):
    """UserWarning on max-iter and on deprecated=True (:445,:543) from the public function."""
    import zfista_amd.proximal_gradient as pg

    d, c, lam = P.make_pdiag(17, seed=3)
    prob = FakeProblem(d, c, lam)
    monkeypatch.setattr(pg, "match_native", lambda *a: prob)
    real = pg._solve_native
    monkeypatch.setattr(pg, "_solve_native", lambda p, x0, o: real(p, x0, o, solver_factory=_factory))
    with pytest.warns(UserWarning, match="Maximum number of iterations reached"):
        res = pg.minimize_proximal_gradient(None, None, None, None, np.zeros(17), lr=0.4, max_iter=2, tol=0.0)
    assert res.nit == 2
    with pytest.warns(UserWarning, match="deprecated option"):
        pg.minimize_proximal_gradient(None, None, None, None, np.zeros(17), lr=0.4, deprecated=True)
    # verbose prints a header and one row per iteration (documented deviation: the reference raises)
    res = pg.minimize_proximal_gradient(None, None, None, None, np.zeros(17), lr=0.4, tol=1e-3, verbose=True)
    assert res.success


# ---------------------------------------------------------------------------
# temporal blocking: chains of S trials per pass (zf_decide_pass, csrc/zf_decide.h)
# ---------------------------------------------------------------------------
ACCEPT = [10.2, -2.0, 1.0, 1.0, 8.0, 0.5, 0, 0]      # see test_decide_accept_reject_and_failure
REJECT = [10.0, -2.0, 1.0, 1.0, 9.5, 0.5, 0, 0]


def _chain_ctl(sub, **kw):
    kw = dict(sub_iters=sub, ring_size=4, cur=0, prev=3, lag=0, pend_status=0, max_backtrack=5) | kw
    return _ctl(**kw)


def _follow(F_old, pack):
    """A pack that is accepted after `pack` was: same model terms, F decreases by 1."""
    p = list(pack)
    p[0], p[4] = F_old - p[3] + 1.2, F_old - p[3] - 1.0   # f_y, f_x  (g_x = p[3])
    return p


def test_decide_pass_commits_a_full_chain():
    c = _chain_ctl(4)
    packs, F = [], 10.0
    for _ in range(4):
        p = _follow(F, ACCEPT)
        packs += p
        F = p[4] + p[3]
    tr = _decide(c, packs)
    assert (c.nit, c.status, c.trial, c.total_trials) == (4, _lib.ZF_RUNNING, 0, 4)
    assert (c.prev, c.cur) == (1, 2)                  # x_{k+3}, x_{k+4}: the two lowest free buffers
    assert (c.lag, c.pend_status) == (0, 0)
    assert np.all(tr[:4, _lib.TR_TRIALS] == 1) and c.F_old == F


def test_decide_pass_first_trial_rejected_is_committed():
    c = _chain_ctl(4)
    _decide(c, REJECT + ACCEPT * 3)
    assert (c.nit, c.trial, c.total_trials, c.lr) == (0, 1, 1, 0.5)
    assert (c.cur, c.prev, c.lag) == (0, 3, 0)


def test_decide_pass_broken_chain_keeps_the_accepted_prefix():
    """A chain that breaks after two acceptances: the two iterations count (nit, F_old, trace rows,
    counters) but their iterates lag - buffers untouched, step sizes remembered for the replay."""
    c = _chain_ctl(4)
    p0 = _follow(10.0, ACCEPT)
    p1 = _follow(p0[4] + p0[3], ACCEPT)
    F2 = p1[4] + p1[3]
    rej = list(REJECT)
    rej[4] = F2 + 5.0                                  # f_x far too large: rejected whatever F_old is
    tr = _decide(c, p0 + p1 + rej + ACCEPT)
    assert (c.nit, c.total_trials, c.trial, c.lr, c.F_old) == (2, 3, 1, 0.5, F2)
    assert (c.cur, c.prev, c.lag, c.pend_status, c.status) == (0, 3, 2, 0, _lib.ZF_RUNNING)
    assert list(c.lag_lr[:2]) == [1.0, 1.0] and list(tr[:2, _lib.TR_LR]) == [1.0, 1.0]
    # next pass: 2 replayed + min(4, 7 - 2) = 4 fresh trials at lr / 2; first one rejected again
    _decide(c, rej + ACCEPT * 3)
    assert (c.nit, c.lag, c.lr, c.trial, c.total_trials) == (2, 2, 0.25, 2, 4)
    # then a chain of 4 that holds: committed as a chain of 2 + 4 trials
    packs, F = [], F2
    for _ in range(4):
        p = _follow(F, ACCEPT)
        packs += p
        F = p[4] + p[3]
    tr = _decide(c, packs)
    assert (c.nit, c.lag, c.lr, c.total_trials, c.trial) == (6, 0, 0.25, 8, 0)
    assert (c.prev, c.cur) == (1, 2)
    assert list(tr[2:6, _lib.TR_TRIALS]) == [3, 1, 1, 1] and list(tr[2:6, _lib.TR_LR]) == [0.25] * 4


def test_decide_pass_lag_bounds_the_fresh_trials():
    """lag + fresh <= 2 S - 1: repeated breaks shorten the fresh part, never the other way round."""
    c = _chain_ctl(2, max_backtrack=50)
    p0 = _follow(10.0, ACCEPT)
    rej = list(REJECT)
    rej[4] = 1e9
    _decide(c, p0 + rej)                               # lag 1
    assert (c.nit, c.lag) == (1, 1)
    F = p0[4] + p0[3]
    p1 = _follow(F, ACCEPT)
    _decide(c, p1 + rej)                               # 2 fresh allowed (1 + 2 <= 3): lag 2
    assert (c.nit, c.lag, list(c.lag_lr[:2])) == (2, 2, [1.0, 0.5])
    F = p1[4] + p1[3]
    p2 = _follow(F, ACCEPT)
    _decide(c, p2 + rej)                               # only 1 fresh trial now: accepted -> chain of 3 commits
    assert (c.nit, c.lag, c.prev, c.cur) == (3, 0, 1, 2)


def test_decide_pass_termination_inside_a_chain():
    c = _chain_ctl(4)
    p0 = _follow(10.0, ACCEPT)
    p1 = _follow(p0[4] + p0[3], ACCEPT)
    p1[5] = 1e-7                                       # err < tol at the second trial
    _decide(c, p0 + p1 + ACCEPT + ACCEPT)
    # converged, but x_k is not stored yet: one materialise-only pass is pending
    assert (c.nit, c.status, c.pend_status, c.lag) == (2, _lib.ZF_RUNNING, _lib.ZF_CONVERGED, 2)
    _decide(c, [0.0] * 32)
    assert (c.nit, c.status, c.pend_status, c.lag, c.prev, c.cur) == (2, _lib.ZF_CONVERGED, 0, 0, 1, 2)
    # max_iter bounds the chain (:539): 3 iterations left -> 3 trials
    c = _chain_ctl(4, max_iter=3)
    packs, F = [], 10.0
    for _ in range(3):
        p = _follow(F, ACCEPT)
        packs += p
        F = p[4] + p[3]
    _decide(c, packs + REJECT)
    assert (c.nit, c.status, c.lag) == (3, _lib.ZF_MAXITER, 0)
    c = _chain_ctl(4, max_iter=1)                      # one-trial chains store one iterate
    _decide(c, _follow(10.0, ACCEPT) + REJECT * 3)
    assert (c.nit, c.status, c.prev, c.cur) == (1, _lib.ZF_MAXITER, 0, 1)
    # a flush request materialises and keeps running
    c = _chain_ctl(4, lag=3, pend_status=_lib.ZF_PEND_FLUSH, nit=3)
    _decide(c, [0.0] * 32)
    assert (c.nit, c.status, c.pend_status, c.lag, c.prev, c.cur) == (3, _lib.ZF_RUNNING, 0, 0, 1, 2)


def test_decide_pass_backtracking_failure_inside_a_chain():
    c = _chain_ctl(2, max_backtrack=2)
    p0 = _follow(10.0, ACCEPT)
    rej = list(REJECT)
    rej[4] = 1e9
    _decide(c, p0 + rej)
    assert (c.nit, c.lag, c.trial, c.status) == (1, 1, 1, _lib.ZF_RUNNING)
    _decide(c, rej + rej)                              # second rejection of the same line search: failure (:306)
    assert (c.nit, c.status, c.pend_status, c.lag) == (1, _lib.ZF_RUNNING, _lib.ZF_BACKTRACK_FAILED, 1)
    _decide(c, [0.0] * 16)                             # x_1 materialised, then the failure stands
    assert (c.nit, c.status, c.lr, c.total_trials, c.lag, c.prev, c.cur) == (1, _lib.ZF_BACKTRACK_FAILED, 0.25, 3, 0, 0, 1)


CHAIN_CASES = [
    dict(lr=0.45, nesterov=True, tol=0.0, max_iter=37),
    dict(lr=0.45, nesterov=True, tol=0.0, max_iter=1),
    dict(lr=0.45, nesterov=True, tol=0.0, max_iter=3),
    dict(lr=64.0, nesterov=True, tol=0.0, max_iter=23),
    dict(lr=64.0, nesterov=False, tol=1e-9, max_iter=500),
    dict(lr=3.0, nesterov=True, decay_rate=0.9, tol=1e-8, max_iter=400),
    dict(lr=0.45, nesterov=True, decay_rate=1.0, tol=1e-7, max_iter=300),
    dict(lr=0.45, nesterov=True, deprecated=True, tol=1e-7, max_iter=300),
    dict(lr=1e6, nesterov=True, max_backtrack_iter=3, tol=0.0, max_iter=50),
    dict(lr=0.45, nesterov=True, tol=1e-12, max_iter=2 * _lib.ZF_RING + 13),
]


def _chain_run(prob, x0, kw, sub, chunk):
    run = NativeRun(prob, x0, BASE | kw | dict(sub_iters=sub), solver_factory=_factory)
    assert run.sub_iters == sub
    rows = [np.zeros((0, _lib.ZF_TRACE_COLS))]
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(chunk))
    c = run.solver.ctl
    return dict(rows=np.concatenate(rows), x=run.solver.get_x(), nit=int(c.nit), status=int(c.status), lr=c.lr,
                F=c.F_old, trials=int(c.total_trials), passes=run.solver.passes)


@pytest.mark.parametrize("case", range(len(CHAIN_CASES)))
def test_chained_passes_do_not_change_results(case):
    """Host driver + decide pass with chains of S trials per pass (the lag / replay logic the
    GPU kernels follow) against one trial per pass: identical traces, iterates and counters."""
    n = 501
    d, c, lam = P.make_pdiag(n, seed=20 + case)
    prob = FakeProblem(d, c, lam)
    x0 = np.random.default_rng(case).standard_normal(n)
    kw = CHAIN_CASES[case]
    ref = _chain_run(prob, x0, kw, 1, 64)
    for sub in (2, 4, 8, 16):
        for chunk in (1, 9):
            r = _chain_run(prob, x0, kw, sub, chunk)
            for key in ("nit", "status", "lr", "F", "trials"):
                assert r[key] == ref[key], (sub, chunk, key)
            assert np.array_equal(r["rows"], ref["rows"]) and np.array_equal(r["x"], ref["x"]), (sub, chunk)
        # a rejection costs at most the pass it was found in, never a second one: the accepted
        # prefix of a broken chain counts (zf_control.lag), + 1 materialise-only pass at the end
        assert r["passes"] <= -(-ref["nit"] // sub) + (ref["trials"] - ref["nit"]) + 2, (sub, r["passes"])
