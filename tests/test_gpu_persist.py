"""GPU: several passes per launch (zf_persist_kernel, opt-in: ZF_PERSIST=1) and the branch-free mid chains (PART 3, 9 ..
15 trials).

On grids the device holds at once (n up to ~2.5e7) consecutive full-chain passes can share ONE launch: the last arriver
of a pass decides it and publishes what the next pass needs past the caches, the other workgroups wait for its sequence
number.  Measured (profiles/r04_persist_*): the in-kernel barrier costs what the kernel boundary costs, so the
persistent kernel is OFF by default - but it is kept correct: everything a per-pass launch produces - trace rows,
iterates, lr / trial sequences, statuses - must come out bit for bit the same, whatever ends a launch early: a rejected trial, a
termination in the middle of a chain, the tail before max_iter.  The tail lengths 9 .. 15 each have a kernel of their
own; S = 16 must equal S = 1 for every max_iter that produces them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_backtrack_iter=100, decay_rate=0.5,
            nesterov=False, nesterov_ratio=(0, 0.25), deprecated=False, return_all=False)


def _pdiag(n, seed=1):
    from oracle import problems_ref as P
    from zfista_amd.problems import DiagQuadL1

    d, c, lam = P.make_pdiag(n, seed=seed)
    return DiagQuadL1(d, c, lam)


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
    out = dict(rows=np.concatenate(rows), x=run.solver.get_x(), nit=int(ctl.nit), status=int(ctl.status), lr=ctl.lr,
               F=ctl.F_old, trials=int(ctl.total_trials), persist=run.solver.persist_counts(),
               launches=run.solver.launch_counts())
    run.solver.close()
    return out


def _same(a, b):
    assert (a["nit"], a["status"], a["lr"], a["F"], a["trials"]) == (b["nit"], b["status"], b["lr"], b["F"], b["trials"])
    assert np.array_equal(a["rows"], b["rows"]) and np.array_equal(a["x"], b["x"])


PERSIST_CASES = [
    (10007, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=200)),                 # clean: long runs of full chains
    (10007, dict(lr=0.45, nesterov=False, tol=0.0, max_iter=333)),
    (300001, dict(lr=0.45, nesterov=True, tol=1e-7, max_iter=5000)),              # terminates inside a chain
    (300001, dict(lr=16.0, nesterov=True, tol=0.0, max_iter=150)),                # rejections first, then chains
    (2_000_003, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=100)),             # one round of 489 workgroups of two tiles
    (10_000_000, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=120)),            # cfg2: ONE round of 489 workgroups x 10 tiles; crosses the noise floor
]


@pytest.mark.parametrize("case", range(len(PERSIST_CASES)))
def test_persistent_passes_equal_per_pass_launches(case, monkeypatch):
    n, opts = PERSIST_CASES[case]
    prob = _pdiag(n, seed=2 + case)
    x0 = np.zeros(n)
    monkeypatch.delenv("ZF_PERSIST", raising=False)
    ref = _run(prob, x0, opts)
    assert ref["persist"] == (0, 0)
    monkeypatch.setenv("ZF_PERSIST", "1")
    for chunk in (64, 5, 2):
        got = _run(prob, x0, opts, chunk=chunk)
        _same(got, ref)
        launches, passes = got["persist"]
        assert launches >= 1 and passes >= 2 * launches, "the persistent kernel was expected to run (a grid of <= 512 workgroups)"
    one = _run(prob, x0, opts, sub=1)     # and both equal the one-trial-per-pass loop
    _same(one, ref)


def test_persistent_kernel_is_not_used_beyond_the_resident_grid(monkeypatch):
    monkeypatch.setenv("ZF_PERSIST", "1")
    n = 30_000_000     # two rounds of workgroups: not co-resident
    r = _run(_pdiag(n, seed=9), np.zeros(n), dict(lr=0.45, nesterov=True, tol=0.0, max_iter=48))
    assert r["persist"] == (0, 0) and r["nit"] == 48


@pytest.mark.parametrize("nesterov", [True, False])
def test_every_tail_length_has_its_kernel_and_changes_nothing(nesterov):
    """max_iter = 17 .. 47: the last S < left < 2 S iterations are shared by two passes of about left / 2 trials and
    fewer than S are one pass - mid chains of every length 9 .. 15 (branch-free, one kernel each).  Exactly predicted
    (chunks of 64) and step by step; against S = 1."""
    n = 10007
    prob = _pdiag(n, seed=21)
    x0 = np.random.default_rng(5).standard_normal(n)
    for max_iter in range(17, 48):
        opts = dict(lr=0.45, nesterov=nesterov, tol=0.0, max_iter=max_iter)
        ref = _run(prob, x0, opts, sub=1)
        for chunk in (64, 1):
            got = _run(prob, x0, opts, chunk=chunk)
            _same(got, ref)
            if chunk == 64:
                steps, kernels = got["launches"]
                # exact prediction: one kernel per step (persistent launches cover several steps with one)
                assert kernels <= steps, (max_iter, steps, kernels)
