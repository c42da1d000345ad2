"""GPU: run-ahead passes (zf_runahead_kernel).

On grids the device holds at once (n up to ~2.5e7) consecutive full-chain passes go alternately to two streams: pass
p + 1 starts while pass p is still being finalised - workgroup j as soon as workgroup j of pass p has stored its
iterates - on the control block the host EXPECTS pass p to leave.  A pass whose predecessor did not go as expected
(a rejected trial, a termination inside the chain) is void.  Everything a solve with one launch per pass on one stream
produces - trace rows, iterates, lr / trial sequences, statuses - must come out bit for bit the same (ZF_RUNAHEAD=0
is that solve), whatever the chunking of the host loop, and equal the one-trial-per-pass loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_backtrack_iter=100, decay_rate=0.5,
            nesterov=False, nesterov_ratio=(0, 0.25), deprecated=False, return_all=False)


def _pdiag(n, seed=1, bounds=None):
    from oracle import problems_ref as P
    from zfista_amd.problems import DiagQuadL1

    d, c, lam = P.make_pdiag(n, seed=seed)
    return DiagQuadL1(d, c, lam) if bounds is None else DiagQuadL1(d, c, lam, bounds=bounds)


def _run(prob, x0, opts, sub=16, chunk=64):
    from zfista_amd import _lib
    from zfista_amd.proximal_gradient import NativeRun

    o = dict(BASE)
    o.update(opts)
    o["sub_iters"] = sub
    run = NativeRun(prob, x0, o)
    rows = [np.zeros((0, _lib.ZF_TRACE_COLS))]
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(chunk))
    ctl = run.solver.ctl
    out = dict(rows=np.concatenate(rows), x=run.solver.get_x(), xp=run.solver.get_x_prev(), nit=int(ctl.nit), status=int(ctl.status),
               lr=ctl.lr, F=ctl.F_old, trials=int(ctl.total_trials), ra=run.solver.runahead_counts(), launches=run.solver.launch_counts())
    run.solver.close()
    return out


def _same(a, b):
    assert (a["nit"], a["status"], a["lr"], a["F"], a["trials"]) == (b["nit"], b["status"], b["lr"], b["F"], b["trials"])
    assert np.array_equal(a["rows"], b["rows"]) and np.array_equal(a["x"], b["x"]) and np.array_equal(a["xp"], b["xp"])


CASES = [
    (10007, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=200), None),            # clean: long runs of full chains
    (10007, dict(lr=0.45, nesterov=False, tol=0.0, max_iter=333), None),
    (4099, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=170), None),             # two workgroups + a remainder
    (4099, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=160), (-0.3, 0.4)),      # a box: its own run-ahead kernels (full chains)
    (300001, dict(lr=0.45, nesterov=True, tol=1e-7, max_iter=5000), None),         # terminates inside a chain: the pass behind is void
    (300001, dict(lr=16.0, nesterov=True, tol=0.0, max_iter=150), None),           # rejections first, then chains
    (1_000_001, dict(lr=0.45, nesterov=True, nesterov_ratio=(0.5, 1 / 16), tol=0.0, max_iter=170), None),
    (2_000_003, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=100), None),        # one round of 489 workgroups of two tiles
    (10_000_000, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=150), None),       # cfg2: 489 workgroups x 10 tiles; crosses the noise floor (chains break)
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_runahead_passes_equal_one_launch_per_pass(case, monkeypatch):
    n, opts, bounds = CASES[case]
    prob = _pdiag(n, seed=2 + case, bounds=bounds)
    x0 = np.zeros(n) if case % 2 == 0 else np.random.default_rng(case).standard_normal(n)
    monkeypatch.setenv("ZF_RUNAHEAD", "0")
    ref = _run(prob, x0, opts)
    assert ref["ra"] == (0, 0)
    monkeypatch.delenv("ZF_RUNAHEAD")
    for chunk in (64, 5, 2, 1):
        got = _run(prob, x0, opts, chunk=chunk)
        _same(got, ref)
        passes, ahead = got["ra"]
        if chunk >= 2:
            assert passes >= 2 and ahead >= 1, "run-ahead passes were expected (a grid of <= 512 workgroups, full chains)"
        else:
            assert ahead == 0   # one step per call: nothing to run ahead of
    if n <= 2_000_003:
        one = _run(prob, x0, opts, sub=1)     # and both equal the one-trial-per-pass loop
        _same(one, ref)


@pytest.mark.parametrize("max_iter,lengths", [(20, (10, 10)), (26, (13, 13)), (45, (16, 15, 14)), (58, (16, 16, 13, 13)),
                                              (31, (16, 15)), (9, (9,)), (75, (16, 16, 16, 14, 13)), (23, (12, 11))])
@pytest.mark.parametrize("nesterov", [True, False])
def test_mid_chains_run_ahead_too(max_iter, lengths, nesterov, monkeypatch):
    """The passes of a tail shared by two passes (K = 20 -> 10 + 10) and the tail itself are branch-free mid chains of
    9 .. 15 trials: as run-ahead passes (zf_runahead_kernel<..., L>) they overlap like the full chains in front of them,
    with the results of one launch per pass, bit for bit.  (n = 5e6: five tiles per workgroup - in the middle of the sizes at
    which a run may START with a mid chain; at every one-round size a mid chain goes on a run that is in flight.)"""
    n = 5_000_011
    prob = _pdiag(n, seed=60 + max_iter)
    opts = dict(lr=0.45, nesterov=nesterov, tol=0.0, max_iter=max_iter)
    x0 = np.random.default_rng(max_iter).standard_normal(n)
    monkeypatch.setenv("ZF_RUNAHEAD", "0")
    ref = _run(prob, x0, opts)
    assert ref["ra"] == (0, 0) and ref["nit"] == max_iter
    clean = ref["trials"] == max_iter   # (no trial rejected: the passes are exactly `lengths`)
    monkeypatch.delenv("ZF_RUNAHEAD")
    got = _run(prob, x0, opts)
    _same(got, ref)
    if clean:
        assert got["ra"] == (len(lengths), len(lengths) - 1), (got["ra"], lengths)   # every pass of the solve, mid chains included
    got = _run(prob, x0, opts, chunk=3)
    _same(got, ref)


def test_where_a_run_starts_with_a_mid_chain(monkeypatch):
    """Small grids: two launches on one stream cost less than the fork and join of two streams - a run does not start
    with a mid chain there (it goes on with one: 16 + 15); the results are the same either way."""
    n = 1_000_001
    prob = _pdiag(n, seed=77)
    for max_iter, want in ((20, (0, 0)), (31, (2, 1))):
        opts = dict(lr=0.45, nesterov=True, tol=0.0, max_iter=max_iter)
        monkeypatch.setenv("ZF_RUNAHEAD", "0")
        ref = _run(prob, np.zeros(n), opts)
        monkeypatch.delenv("ZF_RUNAHEAD")
        got = _run(prob, np.zeros(n), opts)
        _same(got, ref)
        assert got["ra"] == want, (max_iter, got["ra"])


@pytest.mark.parametrize("bounds", [(-0.3, 0.4), (0.0, np.inf)])
def test_clipped_problems_run_ahead(bounds, monkeypatch):
    """Box constraints: the full chains run ahead (their own kernels); tails take the general per-pass body (there are no
    clipped mid chains), so a run of passes ends in front of them."""
    n = 1_000_001
    prob = _pdiag(n, seed=71, bounds=bounds)
    opts = dict(lr=0.45, nesterov=True, tol=0.0, max_iter=106)   # 16 x 5, then 13 + 13
    x0 = np.clip(np.random.default_rng(3).standard_normal(n), bounds[0], min(bounds[1], 10.0))
    monkeypatch.setenv("ZF_RUNAHEAD", "0")
    ref = _run(prob, x0, opts)
    monkeypatch.delenv("ZF_RUNAHEAD")
    for chunk in (64, 3):
        got = _run(prob, x0, opts, chunk=chunk)
        _same(got, ref)
    got = _run(prob, x0, opts)
    assert got["nit"] >= 64 and got["ra"][0] >= 4 and got["ra"][1] >= 3, (got["nit"], got["ra"])   # (a clipped solve may end before max_iter: an iterate that no longer moves)


@pytest.mark.parametrize("seed", range(16))
def test_random_solves_with_and_without_runahead_passes(seed, monkeypatch):
    """A seeded sweep over what decides which passes run ahead - size (tiles per workgroup), the length of the solve (full
    chains, shared tails, single tails), a box, momentum, a step size the line search has to cut first, the host's chunking:
    always the results of one launch per pass, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(150_000, 6_000_000)) | 1
    bounds = None if rng.random() < 0.6 else (-float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.2, 0.8)))
    opts = dict(lr=float(rng.choice([0.45, 0.45, 0.9, 8.0])), nesterov=bool(rng.random() < 0.75), tol=0.0,
                max_iter=int(rng.integers(9, 140)))
    if rng.random() < 0.3:
        opts["nesterov_ratio"] = (0.5, 1 / 16)
    prob = _pdiag(n, seed=500 + seed, bounds=bounds)
    x0 = rng.standard_normal(n) if bounds is None else np.clip(rng.standard_normal(n), *bounds)
    chunk = int(rng.choice([64, 7, 3, 2]))
    monkeypatch.setenv("ZF_RUNAHEAD", "0")
    ref = _run(prob, x0, opts, chunk=chunk)
    monkeypatch.delenv("ZF_RUNAHEAD")
    got = _run(prob, x0, opts, chunk=chunk)
    _same(got, ref)
    # (a clean solve of two FULL chains or more: something ran ahead.  Clipped problems have no mid chains: 48 iterations are
    #  16 + 16 + 16, 41 are 16 + 13 + 12 - one full chain, and the tail takes the general per-pass body)
    if opts["max_iter"] >= (32 if bounds is None else 48) and ref["trials"] == ref["nit"] and ref["nit"] == opts["max_iter"] and chunk >= 2:
        assert got["ra"][1] >= 1, (n, opts, bounds, got["ra"])


def test_runahead_is_not_used_beyond_the_resident_grid():
    n = 30_000_000     # two rounds of workgroups
    r = _run(_pdiag(n, seed=9), np.zeros(n), dict(lr=0.45, nesterov=True, tol=0.0, max_iter=48))
    assert r["ra"] == (0, 0) and r["nit"] == 48


def test_a_wait_that_gives_up_voids_the_pass_and_the_solve_recovers(monkeypatch):
    """ZF_RUNAHEAD_SPIN_LIMIT=0: every workgroup of a pass launched behind a pass in flight gives up at once - the pass
    contributes rows of zeros and must be void; the solve continues from the control block the next poll reads."""
    n = 1_000_001
    prob = _pdiag(n, seed=31)
    opts = dict(lr=0.45, nesterov=True, tol=0.0, max_iter=160)
    monkeypatch.setenv("ZF_RUNAHEAD", "0")
    ref = _run(prob, np.zeros(n), opts)
    monkeypatch.delenv("ZF_RUNAHEAD")
    monkeypatch.setenv("ZF_RUNAHEAD_SPIN_LIMIT", "0")
    got = _run(prob, np.zeros(n), opts, chunk=8)
    _same(got, ref)


def test_snapshot_and_resume_across_runahead_passes():
    from zfista_amd import _lib
    from zfista_amd.proximal_gradient import NativeRun

    n = 500_003
    prob = _pdiag(n, seed=41)
    o = dict(BASE, lr=0.45, nesterov=True, tol=0.0, max_iter=192, sub_iters=16)
    run = NativeRun(prob, np.zeros(n), o)
    while run.status == _lib.ZF_RUNNING:
        run.advance(64)
    want_x, want_F = run.solver.get_x(), run.solver.ctl.F_old
    run.solver.close()
    run = NativeRun(prob, np.zeros(n), o)
    run.advance(5)
    snap = run.snapshot()
    run.solver.close()
    run = NativeRun.from_snapshot(prob, snap, o)
    while run.status == _lib.ZF_RUNNING:
        run.advance(3)
    assert run.solver.runahead_counts()[1] >= 1
    assert np.array_equal(run.solver.get_x(), want_x) and run.solver.ctl.F_old == want_F
    run.solver.close()


def test_the_step_counter_wraps_under_run_ahead_passes(monkeypatch):
    """Sequence numbers are compared (flags, done / good words); the counter wraps at 0x7ffffff0 -> 1: the run of passes
    is joined there and the words are cleared before the second stream may look at them."""
    n = 300_001
    prob = _pdiag(n, seed=51)
    opts = dict(lr=0.45, nesterov=True, tol=0.0, max_iter=240)
    ref = _run(prob, np.zeros(n), opts)
    monkeypatch.setenv("ZF_PASS_SEQ_START", str(0x7ffffff0 - 6))
    for chunk in (64, 3):
        got = _run(prob, np.zeros(n), opts, chunk=chunk)
        _same(got, ref)
        assert got["ra"][1] >= 4
