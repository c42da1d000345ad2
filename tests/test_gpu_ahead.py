"""GPU: passes AHEAD of their predecessor's decision at kernel granularity (zf_trial_kernel<..., AHEAD>, zf_tail_kernel,
zf_decide_ahead_kernel; csrc/zf_solver.hip: zf_launch_ahead), and what run-ahead passes report about themselves.

A sharded solve through the library's communicator runs the trial kernels of consecutive, exactly predicted full / mid
chains back to back on the solver's stream - each on the head the host expects - while rows -> packs -> all-gather ->
decide of the pass before run on a second stream; the trial kernel two passes on is launched behind an event of that
decide step and leaves at once unless the pass went as expected.  A pass on a head that did not come true is VOID.
Everything the sequence trial -> all-gather -> decide -> trial (ZF_AHEAD=0) produces - trace rows, iterates, lr / trial
sequences, statuses - must come out bit for bit the same, on every rank, whatever the chunking of the host loop.
Transports: a real 1-rank RCCL communicator (ncclAllGather on the second stream) and 2 / 3 thread ranks on this GPU."""
import threading
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_backtrack_iter=100, decay_rate=0.5,
            nesterov=False, nesterov_ratio=(0, 0.25), deprecated=False, return_all=False)


def _data(n, seed):
    from oracle import problems_ref as P

    return P.make_pdiag(n, seed=seed)


def _solve(prob, x0, opts, chunk=64, sub=16):
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
               lr=ctl.lr, F=ctl.F_old, trials=int(ctl.total_trials), report=run.solver.ahead_report(),
               launches=run.solver.launch_counts())
    run.solver.close()
    return out


def _same(a, b):
    assert (a["nit"], a["status"], a["lr"], a["F"], a["trials"]) == (b["nit"], b["status"], b["lr"], b["F"], b["trials"])
    assert np.array_equal(a["rows"], b["rows"]) and np.array_equal(a["x"], b["x"]) and np.array_equal(a["xp"], b["xp"])


def _sharded(world, n, seed, opts, chunk, bounds=None, x0=None, acceptance=None):
    """The solve with x split over `world` ranks of a library communicator; returns the per-rank results."""
    import torch

    from zfista_amd.comm import LibComm
    from zfista_amd.problems import DiagQuadL1

    d, c, lam = _data(n, seed)
    x0 = np.zeros(n) if x0 is None else x0
    o = dict(opts)
    if acceptance:
        o["acceptance"] = acceptance
    comms = LibComm.local_group(world, cap_doubles=4096) if world > 1 else [LibComm(0, 1, LibComm.new_unique_id())]
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            lo, hi = r * n // world, (r + 1) * n // world
            with torch.cuda.stream(torch.cuda.Stream()):
                prob = DiagQuadL1(d[lo:hi], c[lo:hi], lam, bounds=bounds, group=comms[r])
                out[r] = _solve(prob, x0[lo:hi], o, chunk=chunk)
                torch.cuda.current_stream().synchronize()
        except Exception as exc:   # pragma: no cover - reported below
            errs.append(exc)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    for c_ in comms:
        c_.close()
    assert not errs, errs
    assert all(o_ is not None for o_ in out), "a rank thread did not finish"
    return out


CASES = [
    # n, options, box
    (30011, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=200), None),             # clean: long runs of full chains
    (30011, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=16 * 4 + 20), None),     # ... and a shared tail of two mid chains
    (4099 * 3, dict(lr=0.45, nesterov=False, tol=0.0, max_iter=150), (-0.3, 0.4)),  # a box: passes ahead take clipped problems too
    (300001, dict(lr=0.45, nesterov=True, tol=1e-7, max_iter=5000), None),          # terminates inside a chain: the passes behind are void
    (300001, dict(lr=16.0, nesterov=True, tol=0.0, max_iter=150), None),            # rejections first, then chains
    (3_000_001, dict(lr=0.45, nesterov=True, tol=0.0, max_iter=170), None),         # crosses the noise floor per shard size (chains break)
]


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_passes_ahead_through_the_communicator_equal_sequential_passes(case, world, monkeypatch):
    n, opts, bounds = CASES[case]
    x0 = np.zeros(n) if case % 2 == 0 else np.random.default_rng(case).standard_normal(n)
    monkeypatch.setenv("ZF_RUNAHEAD_SHARDED", "0")   # (these grids fit the device: by default they take sharded run-ahead passes, below)
    monkeypatch.setenv("ZF_AHEAD", "0")
    ref = _sharded(world, n, 2 + case, opts, 64, bounds, x0)
    assert all(r["report"]["ahead"] == 0 for r in ref)
    monkeypatch.delenv("ZF_AHEAD")
    for chunk in (64, 5, 2, 1):
        got = _sharded(world, n, 2 + case, opts, chunk, bounds, x0)
        for r in range(world):
            _same(got[r], ref[r])
            rep = got[r]["report"]
            assert rep == got[0]["report"], "every rank launches the same passes"
            if chunk >= 2:
                assert rep["ahead"] >= 2, "passes ahead were expected (exactly predicted full / mid chains through a communicator)"
            assert rep["timeouts"] == 0 and rep["runahead"] == 0   # (nothing waits inside a kernel on this path)
        if case in (3, 4, 5) and chunk == 64:
            assert got[0]["report"]["ahead_void"] >= 1, "a chain broke inside a chunk: the passes behind it are void and counted"
    # all ranks bit-identical in their scalars, and the shards are the slices of the unsharded solve
    for r in range(1, world):
        assert np.array_equal(ref[r]["rows"], ref[0]["rows"]) and ref[r]["F"] == ref[0]["F"]
    from zfista_amd.problems import DiagQuadL1

    d, c, lam = _data(n, 2 + case)
    one = _solve(DiagQuadL1(d, c, lam, bounds=bounds), x0, opts)
    if world == 1:
        _same(ref[0], one)
    else:
        assert one["nit"] == ref[0]["nit"] and one["status"] == ref[0]["status"]
        x = np.concatenate([r["x"] for r in ref])
        if one["trials"] == one["nit"] and ref[0]["trials"] == ref[0]["nit"]:
            # no trial rejected on either layout: the iterate arithmetic is elementwise - the shards ARE the slices
            assert np.array_equal(x, one["x"])
        else:
            # rejections at the resolution limit of the acceptance test depend on the summation order, hence on the rank
            # layout: the solves agree as far as the test resolves
            assert np.linalg.norm(x - one["x"]) <= 1e-5 * max(1.0, np.linalg.norm(one["x"]))


def test_passes_ahead_on_an_unsharded_grid_of_several_rounds(monkeypatch):
    """ZF_AHEAD_UNSHARDED=1: the same scheme without a communicator, for grids the run-ahead kernel does not take (more
    workgroups than the device holds at once - forced here with one tile per workgroup at a size the suite can afford)."""
    from zfista_amd.problems import DiagQuadL1

    n = 2_000_003
    d, c, lam = _data(n, 77)
    monkeypatch.setenv("ZF_TILES_PER_WG", "1")   # 977 workgroups: not a one-round grid
    for opts in (dict(lr=0.45, nesterov=True, tol=0.0, max_iter=200), dict(lr=8.0, nesterov=True, tol=0.0, max_iter=90),
                 dict(lr=0.45, nesterov=True, tol=2e-7, max_iter=4000)):
        monkeypatch.setenv("ZF_AHEAD_UNSHARDED", "0")
        ref = _solve(DiagQuadL1(d, c, lam), np.zeros(n), opts)
        assert ref["report"]["ahead"] == 0 and ref["report"]["runahead"] == 0
        monkeypatch.setenv("ZF_AHEAD_UNSHARDED", "1")
        for chunk in (64, 3):
            got = _solve(DiagQuadL1(d, c, lam), np.zeros(n), opts, chunk=chunk)
            _same(got, ref)
            assert got["report"]["ahead"] >= 2 and got["report"]["runahead"] == 0


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_sharded_run_ahead_passes_equal_sequential_passes(case, world, monkeypatch):
    """One-round grids behind a communicator (round 5, second half): the run-ahead kernels with packs instead of a decision,
    all-gather + decide of pass p on a third stream beside the workgroups of pass p + 1.  The results of one pass at a time,
    bit for bit, whatever the chunking; every rank launches the same passes.  World 1 = the 1-rank RCCL communicator; thread
    ranks (several ranks of ONE device - by default they keep to passes ahead) only with grids of a few workgroups, all
    of which the device holds at once."""
    n, opts, bounds = CASES[case]
    if world > 1 and n > 100_000:
        pytest.skip("thread ranks share one device: their run-ahead passes would wait for workgroups that have no slot")
    x0 = np.zeros(n) if case % 2 == 0 else np.random.default_rng(case).standard_normal(n)
    monkeypatch.setenv("ZF_RUNAHEAD_SHARDED", "0")
    monkeypatch.setenv("ZF_AHEAD", "0")
    ref = _sharded(world, n, 2 + case, opts, 64, bounds, x0)
    assert all(r["report"]["ahead"] == 0 and r["report"]["runahead"] == 0 for r in ref)
    monkeypatch.delenv("ZF_AHEAD")
    monkeypatch.setenv("ZF_RUNAHEAD_SHARDED", "1")
    for chunk in (64, 5, 2, 1):
        got = _sharded(world, n, 2 + case, opts, chunk, bounds, x0)
        for r in range(world):
            _same(got[r], ref[r])
            rep = got[r]["report"]
            assert rep == got[0]["report"], "every rank launches the same passes"
            if chunk >= 2:
                assert rep["runahead"] >= 2 and rep["runahead_overlapped"] >= 1, rep
            assert rep["timeouts"] == 0 and not rep["runahead_off"]
        if case in (3, 4, 5) and chunk == 64:
            rep = got[0]["report"]
            assert rep["ahead_void"] + rep["void"] >= 1, "a chain broke inside a chunk: the passes behind it are void and counted"


@pytest.mark.parametrize("bounds", [None, (-0.3, 0.4)])
def test_the_two_schemes_mixed_in_one_chunk_start_their_runs_on_the_real_block(bounds, monkeypatch):
    """A run that starts in the middle of a chunk starts on a PREDICTED head.  With both schemes on (unsharded: run-ahead
    passes for full chains, passes ahead - ZF_AHEAD_UNSHARDED=1 - for what those do not take: the tail, and every mid chain
    of a clipped problem) and a step size the line search cuts several times inside the first chunk, the first kernel of every
    run has to find out from the block that its head did not come true - or it writes its iterates over the real x_k
    (found with the sharded run-ahead passes; the same hole, closed the same way)."""
    n = 300_001
    d, c, lam = _data(n, 77)
    from zfista_amd.problems import DiagQuadL1

    for opts in (dict(lr=16.0, nesterov=True, tol=0.0, max_iter=150), dict(lr=0.45, nesterov=True, tol=0.0, max_iter=16 * 5 + 26)):
        monkeypatch.setenv("ZF_RUNAHEAD", "0")
        monkeypatch.setenv("ZF_AHEAD_UNSHARDED", "0")
        ref = _solve(DiagQuadL1(d, c, lam, bounds=bounds), np.zeros(n), opts)
        monkeypatch.delenv("ZF_RUNAHEAD")
        monkeypatch.setenv("ZF_AHEAD_UNSHARDED", "1")
        for chunk in (64, 7, 2):
            got = _solve(DiagQuadL1(d, c, lam, bounds=bounds), np.zeros(n), opts, chunk=chunk)
            _same(got, ref)
        got = _solve(DiagQuadL1(d, c, lam, bounds=bounds), np.zeros(n), opts, chunk=64)
        assert got["report"]["runahead"] >= 1, got["report"]


def test_sharded_run_ahead_passes_at_the_size_of_cfg2(monkeypatch):
    """n = 10^7 behind the 1-rank RCCL communicator: one round of 489 workgroups of ten tiles, both sharded schemes, against
    the unsharded solve - bit for bit (150 iterations: across the noise floor of this size, chains break)."""
    n = 10_000_000
    opts = dict(lr=0.45, nesterov=True, tol=0.0, max_iter=150)
    d, c, lam = _data(n, 5)
    from zfista_amd.problems import DiagQuadL1

    one = _solve(DiagQuadL1(d, c, lam), np.zeros(n), opts)
    for scheme in ("1", "0"):
        monkeypatch.setenv("ZF_RUNAHEAD_SHARDED", scheme)
        got = _sharded(1, n, 5, opts, 64, None, np.zeros(n))[0]
        _same(got, one)
        assert got["report"]["runahead" if scheme == "1" else "ahead"] >= 2, got["report"]


def test_waits_that_give_up_are_counted_and_switch_run_ahead_off(monkeypatch):
    """The advisor's finding of round 4: a run-ahead wait that times out cost its whole limit, voided the pass - and was
    invisible.  ZF_RUNAHEAD_SPIN_LIMIT=0 makes every workgroup of a pass launched behind a pass in flight give up at once (without looking: what it would find depends on the box).
    The device counts it, the next poll reads the count, the solver launches no further run-ahead passes, and the
    result of minimize_proximal_gradient says so - with every bit of the solve as it is without run-ahead passes."""
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import DiagQuadL1

    n = 1_000_001
    d, c, lam = _data(n, 31)
    opts = dict(lr=0.45, nesterov=True, tol=0.0, max_iter=320)
    monkeypatch.setenv("ZF_RUNAHEAD", "0")
    ref = _solve(DiagQuadL1(d, c, lam), np.zeros(n), opts)
    monkeypatch.delenv("ZF_RUNAHEAD")
    clean = _solve(DiagQuadL1(d, c, lam), np.zeros(n), opts, chunk=4)
    _same(clean, ref)
    assert clean["report"]["timeouts"] == 0 and not clean["report"]["runahead_off"] and clean["report"]["runahead_overlapped"] >= 2
    monkeypatch.setenv("ZF_RUNAHEAD_SPIN_LIMIT", "0")
    got = _solve(DiagQuadL1(d, c, lam), np.zeros(n), opts, chunk=4)
    _same(got, ref)
    rep = got["report"]
    assert rep["timeouts"] >= 1 and rep["void"] >= 1 and rep["runahead_off"]
    # switched off at the first poll that saw a timeout: the 20 passes of the solve are not all run-ahead launches
    assert rep["runahead"] < clean["report"]["runahead"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize_proximal_gradient(*DiagQuadL1(d, c, lam).callbacks(), np.zeros(n), **dict(opts, max_iter=400))
    assert "switched off" in res["runahead"] and np.isfinite(res.fun)


def test_two_unsharded_solvers_share_one_gpu(monkeypatch):
    """Two solvers of one-round grids, each with its own pair of streams, enqueueing passes alternately on one device:
    the co-residency the run-ahead kernel counts on is not guaranteed then.  Whatever happens - overlap, waits that
    give up, void passes - both solves come out bit for bit as alone, and the counters tell what happened."""
    import torch

    from zfista_amd import _lib
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    n = 6_000_001   # 489 workgroups each: two solvers want 4 x 489 resident workgroups where the device holds 512
    opts = dict(BASE, lr=0.45, nesterov=True, tol=0.0, max_iter=160, sub_iters=16)
    probs = [DiagQuadL1(*_data(n, 61 + k)) for k in range(2)]
    alone = [_solve(p, np.zeros(n), opts) for p in probs]
    monkeypatch.setenv("ZF_RUNAHEAD_SPIN_LIMIT", "64")   # (a wait that cannot be satisfied soon gives up after ~0.1 ms, not 15)
    streams = [torch.cuda.Stream() for _ in range(2)]
    runs = []
    for p, st in zip(probs, streams):
        with torch.cuda.stream(st):
            runs.append(NativeRun(p, np.zeros(n), opts))
    rows = [[], []]
    while any(r.status == _lib.ZF_RUNNING for r in runs):
        for r, st in zip(runs, streams):      # both enqueue a chunk before either is polled
            if r.status == _lib.ZF_RUNNING:
                with torch.cuda.stream(st):
                    r.enqueue_only(4)
        for k, (r, st) in enumerate(zip(runs, streams)):
            if r.status == _lib.ZF_RUNNING:
                with torch.cuda.stream(st):
                    rows[k].append(r.collect())
    for k, r in enumerate(runs):
        rep = r.solver.ahead_report()
        assert r.solver.ctl.nit == alone[k]["nit"] and r.solver.ctl.F_old == alone[k]["F"]
        assert np.array_equal(np.concatenate(rows[k]), alone[k]["rows"]) and np.array_equal(r.solver.get_x(), alone[k]["x"])
        assert rep["runahead"] >= 1
        assert rep["runahead_off"] == (rep["timeouts"] > 0)
        r.solver.close()
