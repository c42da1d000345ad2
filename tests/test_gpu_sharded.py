"""GPU: the device code of the sharded (N > 1) path on ONE GPU.

Two solver objects (rank 0 / rank 1 shards of the same decision vector) run in
lockstep in one process; the RCCL all-gather of the per-trial packs is emulated
by device-to-device copies (two ranks cannot share one GPU under RCCL).  Every
kernel of the N > 1 path runs: split trial/finalize, decide over world packs in
rank order, sharded initialisation.  The real collective is covered by
tests/test_dist_gloo.py (gloo, world_size 2) through the same gather_packs()."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,shape", [(2, (300, 1000)), (3, (129, 301)), (4, (64, 4096))])
@pytest.mark.parametrize("nesterov", [False, True])
def test_sharded_least_squares_lockstep(world, shape, nesterov):
    """Column-sharded LASSO: rank p holds A_p, x_p; per trial the ranks
    exchange the m-vector A_p x_p (C2) and the scalar pack (C1).  Two..four solver objects
    in lockstep on one GPU, collectives emulated by D2D copies; compared with the oracle on
    the unsharded problem (tolerance 1e-10: the row sums are added in a different order)."""
    import torch

    from conftest import rel_err
    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.engine import DeviceSolver, momentum_factors

    m, n = shape
    A, b, lam = P.make_plasso(m, n, seed=6, n_informative=max(2, n // 10))
    K = 40
    kw = dict(lr=1.0, nesterov=nesterov, tol=0.0, max_iter=K)
    opts = dict(lr=1.0, tol=0.0, tol_internal=1e-12, decay_rate=0.5, max_iter=K, max_backtrack_iter=100,
                nesterov=int(nesterov), deprecated=0)
    bounds = [r * n // world for r in range(world + 1)]
    bd = torch.from_numpy(b).cuda()
    solvers, keep = [], [bd]
    for r in range(world):
        Ap = torch.from_numpy(np.ascontiguousarray(A[:, bounds[r]:bounds[r + 1]])).cuda()
        x0 = torch.zeros(Ap.shape[1], dtype=torch.float64, device="cuda")
        keep += [Ap, x0]
        fields = dict(kind=_lib.ZF_PROBLEM_LEAST_SQUARES_L1, world=world, rank=r, n=Ap.shape[1], m_rows=m,
                      d=None, c=None, A=Ap.data_ptr(), b=bd.data_ptr(), scale=0.5, lam=lam,
                      box_lo=-np.inf, box_hi=np.inf)
        s = DeviceSolver(fields, opts, keepalive=(Ap, bd))
        s.init_begin(x0.data_ptr())
        solvers.append(s)

    def exchange_svec():
        allp = torch.cat([s._s_part for s in solvers])
        for s in solvers:
            s._s_all.copy_(allp)

    def exchange_pack():
        allp = torch.cat([s._pack_local for s in solvers])
        for s in solvers:
            s._pack_all.copy_(allp)

    exchange_svec()
    for s in solvers:
        s.init_finish()
    exchange_pack()
    for s in solvers:
        s.init_commit()
    betas = np.concatenate([[0.0], momentum_factors(K, (0, 0.25))[0]])
    if nesterov:
        for s in solvers:
            s.set_beta(0, betas[:K + 1])
    funs, lrs = [], []
    status, seen, guard = _lib.ZF_RUNNING, 0, 0
    while status == _lib.ZF_RUNNING and guard < 50:
        guard += 1
        for _ in range(16):
            for s in solvers:
                s.enqueue_trial()
            exchange_svec()
            for s in solvers:
                s.trial_finish()
            exchange_pack()
            for s in solvers:
                s.enqueue_decide()
        ctls = [s.poll() for s in solvers]
        c0, t0 = ctls[0]
        for ck, tk in ctls[1:]:
            assert (ck.nit, ck.status, ck.lr, ck.cur) == (c0.nit, c0.status, c0.lr, c0.cur)
            assert np.array_equal(tk, t0)
        rows = t0[np.arange(seen, c0.nit) % _lib.ZF_RING]
        funs += list(rows[:, _lib.TR_F])
        lrs += list(rows[:, _lib.TR_LR])
        seen, status = int(c0.nit), int(c0.status)
    x = np.concatenate([s.get_x() for s in solvers])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.LeastSquaresL1Ref(A, b, lam).callbacks(), np.zeros(n),
                                                 return_all=True, **kw)
    assert seen == exp.nit == K
    assert np.array_equal(np.asarray(lrs), np.asarray(exp.alllrs))
    assert rel_err(x, exp.x) <= 1e-10
    np.testing.assert_allclose(funs, exp.allfuns[1:], rtol=1e-10)
    for s in solvers:
        s.close()


@pytest.mark.parametrize("kw", [
    dict(lr=0.45, nesterov=True, tol=0.0, max_iter=40),
    dict(lr=4.0, nesterov=True, tol=1e-6, max_iter=300),
    dict(lr=3.0, nesterov=False, tol=1e-5, max_iter=300),
])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_lockstep_matches_oracle(kw, world):
    import torch

    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.engine import DeviceSolver, momentum_factors

    n = 10007
    d, c, lam = P.make_pdiag(n, seed=1)
    opts = dict(lr=kw["lr"], tol=kw["tol"], tol_internal=1e-12, decay_rate=0.5, max_iter=kw["max_iter"],
                max_backtrack_iter=100, nesterov=int(kw["nesterov"]), deprecated=0)
    bounds = [r * n // world for r in range(world + 1)]
    solvers, keep = [], []
    for r in range(world):
        dd = torch.from_numpy(d[bounds[r]:bounds[r + 1]].copy()).cuda()
        cc = torch.from_numpy(c[bounds[r]:bounds[r + 1]].copy()).cuda()
        x0 = torch.zeros(dd.numel(), dtype=torch.float64, device="cuda")
        keep += [dd, cc, x0]
        fields = dict(kind=_lib.ZF_PROBLEM_DIAG_QUAD_L1, world=world, rank=r, n=dd.numel(), m_rows=0,
                      d=dd.data_ptr(), c=cc.data_ptr(), A=None, b=None, scale=0.5, lam=lam,
                      box_lo=-np.inf, box_hi=np.inf)
        s = DeviceSolver(fields, opts, keepalive=(dd, cc))
        s.init_begin(x0.data_ptr())
        solvers.append(s)

    def exchange():
        allp = torch.cat([s._pack_local for s in solvers])
        for s in solvers:
            s._pack_all.copy_(allp)

    exchange()
    for s in solvers:
        s.init_commit()
    betas = np.concatenate([[0.0], momentum_factors(kw["max_iter"], (0, 0.25))[0]])
    filled = 0
    errs, funs = [], []
    status = _lib.ZF_RUNNING
    seen = 0
    while status == _lib.ZF_RUNNING:
        chunk = 50
        ahead = chunk * solvers[0].sub_iters   # a pass may accept up to sub_iters iterations
        # factors for accepted counts <= seen + ahead: the last decide resolves the next trial's
        if kw["nesterov"] and seen + ahead + 1 > filled:
            for s in solvers:
                s.set_beta(filled, betas[filled:seen + ahead + 1])
            filled = min(seen + ahead + 1, betas.size)
        for _ in range(chunk):
            for s in solvers:
                s.enqueue_trial()
            exchange()
            for s in solvers:
                s.enqueue_decide()
        ctls = [s.poll() for s in solvers]
        c0, t0 = ctls[0]
        for ck, tk in ctls[1:]:
            assert (ck.nit, ck.status, ck.lr, ck.cur, ck.total_trials) == (c0.nit, c0.status, c0.lr, c0.cur,
                                                                          c0.total_trials)
            assert np.array_equal(tk, t0), "ranks must take bitwise-identical decisions"
        rows = t0[np.arange(seen, c0.nit) % _lib.ZF_RING]
        errs += list(rows[:, _lib.TR_ERR])
        funs += list(rows[:, _lib.TR_F])
        seen, status = int(c0.nit), int(c0.status)
    x = np.concatenate([s.get_x() for s in solvers])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n),
                                                 return_all=True, **kw)
    assert seen == exp.nit
    assert np.array_equal(x, exp.x)
    np.testing.assert_allclose(errs, exp.allerrs, rtol=1e-10, atol=0)
    np.testing.assert_allclose(funs, exp.allfuns[1:], rtol=1e-10)
    for s in solvers:
        s.close()


# ---------------------------------------------------------------------------
# sharded multi-objective: one exchange of 2m+2 raw totals per dual evaluation
# ---------------------------------------------------------------------------
class _ThreadGroup:
    """Stands in for a torch.distributed group: `world` Python threads of this process, one per
    rank, exchange through a barrier.  (Engines are driven from the host, synchronously, so ranks
    cannot be stepped in lockstep from one thread as the single-objective kernels are above.)"""

    def __init__(self, rank, world, shared):
        self.rank, self.world, self._s = rank, world, shared

    def all_gather_host(self, arr):
        s = self._s
        s["slots"][self.rank] = arr
        s["barrier"].wait()
        parts = [np.array(a) for a in s["slots"]]
        s["barrier"].wait()
        return parts


@pytest.mark.parametrize("case", ["jos1_l1", "fds_l1", "fds_box"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_multiobjective_threads(case, world):
    import threading

    # m = 3: the library's own simplex solver (a dozen dual evaluations per trial); SciPy's
    # trust-constr needs 1e3 - 1e5 of them here, each an exchange between the rank threads
    solver = "native" if case.startswith("fds") else "scipy"

    from zfista_amd.problems import FDS, JOS1

    # FDS: f_1 ~ sum k (x_k - k)^4 / n^2 grows like n^3; its cancellation noise reaches the dual
    # gradient and SciPy's trust-constr path (tests/test_gpu_multiobjective.py uses n <= 100 too)
    n = 1003 if case.startswith("jos1") else 103
    mk = {
        "jos1_l1": lambda g: JOS1(n, l1_ratios=np.arange(1, 3) / n, l1_shifts=[0, 1], group=g),
        "fds_l1": lambda g: FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0, 1, 2], group=g),
        "fds_box": lambda g: FDS(n, bounds=(-1.5, 1.8), group=g),
    }[case]
    kw = dict(lr=1.0 if case.startswith("jos1") else 1e-3, nesterov=True, tol=1e-9, max_iter=8, dual_solver=solver)
    x0 = np.random.default_rng(3).uniform(-1, 1, n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = mk(None).minimize_proximal_gradient(x0, return_all=True, **kw)

    shared = dict(slots=[None] * world, barrier=threading.Barrier(world))
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            prob = mk(_ThreadGroup(r, world, shared))
            lo, hi = prob.shard_bounds()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = prob.minimize_proximal_gradient(x0[lo:hi], return_all=True, **kw)
            out[r] = (res, prob._engine().n_exchanges, prob._engine().n_dual_evals)
        except Exception as exc:   # pragma: no cover - reported below
            errs.append(exc)
            shared["barrier"].abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs
    res0 = out[0][0]
    for res, n_ex, n_dual in out:
        assert res.nit == res0.nit == full.nit and res.status == full.status
        assert np.array_equal(np.asarray(res.allerrs), np.asarray(res0.allerrs)), "ranks must agree bit for bit"
        assert np.array_equal(np.stack(res.allfuns), np.stack(res0.allfuns))
        assert n_ex >= n_dual > 0            # one exchange per dual evaluation (+ f, g, recovery)
    x = np.concatenate([o[0].x for o in out])
    tol = 1e-7 if case.startswith("jos1") else 2e-5
    assert np.linalg.norm(x - full.x) <= tol * max(1.0, np.linalg.norm(full.x))
    np.testing.assert_allclose(np.stack(res0.allfuns), np.stack(full.allfuns), rtol=1e-6 if case.startswith("jos1") else 1e-4)
