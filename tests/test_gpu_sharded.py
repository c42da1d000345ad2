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
        if kw["nesterov"] and seen + chunk > filled:
            for s in solvers:
                s.set_beta(filled, betas[filled:seen + chunk])
            filled = seen + chunk
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
