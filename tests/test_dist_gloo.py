"""CPU, world_size 2 over gloo: the sharded (N > 1) path of the host driver.

Each rank owns a contiguous block of x, d, c (SURVEY 8e, P-diag); per trial the
ranks exchange ONE packed all-gather (zfista_amd.engine.gather_packs - the same
function the GPU path calls on RCCL) and each runs the product's decide step on
the gathered packs.  Checks: both ranks take identical decisions, the
concatenated iterates equal the single-process oracle on the full vector
(bit-exact: the element recursion never sees a reduced sum), and the scalar
traces agree to 1e-12."""
import os
import sys
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _init_gloo(dist, rank, world, outdir):
    """File rendezvous in the test's own tmp_path: no port is picked, so none can be taken by another
    process (or by a sibling rank's retrying connect) before rank 0 listens on it."""
    dist.init_process_group("gloo", init_method=f"file://{outdir}/rdzv", rank=rank, world_size=world)


def _worker(rank, world, n, kw, outdir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    from fake_engine import FakeProblem, FakeSolver
    from oracle import problems_ref as P
    from zfista_amd.proximal_gradient import _solve_native

    _init_gloo(dist, rank, world, outdir)
    d, c, lam = P.make_pdiag(n, seed=1)
    lo, hi = rank * n // world, (rank + 1) * n // world
    prob = FakeProblem(d[lo:hi], c[lo:hi], lam, group=dist.group.WORLD, world=world, rank=rank)
    base = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_iter_internal=100000,
                max_backtrack_iter=100, warm_start=False, decay_rate=0.5, nesterov=False,
                nesterov_ratio=(0, 0.25), return_all=True, verbose=False, deprecated=False)
    res, status = _solve_native(prob, np.zeros(hi - lo), base | kw,
                                solver_factory=lambda f, o, p, x0: FakeSolver(f, o, p, x0))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), x=res.x, nit=res.nit, status=status,
             allerrs=np.asarray(res.allerrs), allfuns=np.asarray(res.allfuns),
             vecs=np.stack(res.allvecs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kw", [
    dict(lr=0.45, nesterov=True, tol=0.0, max_iter=40),
    dict(lr=4.0, nesterov=True, tol=1e-6, max_iter=500),      # backtracking: lr decisions must agree
    dict(lr=0.45, nesterov=False, tol=1e-4, max_iter=500),
])
def test_sharded_pdiag_world2(tmp_path, kw):
    import torch.multiprocessing as mp

    from oracle import cpu_ref, problems_ref as P

    n, world = 10007, 2
    mp.spawn(_worker, args=(world, n, kw, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    d, c, lam = P.make_pdiag(n, seed=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n),
                                                 return_all=True, **kw)
    assert int(r[0]["nit"]) == int(r[1]["nit"]) == exp.nit
    assert int(r[0]["status"]) == int(r[1]["status"])
    assert np.array_equal(r[0]["allerrs"], r[1]["allerrs"]) and np.array_equal(r[0]["allfuns"], r[1]["allfuns"])
    full = np.concatenate([r[0]["vecs"], r[1]["vecs"]], axis=1)
    assert np.array_equal(full, np.stack(exp.allvecs))
    np.testing.assert_allclose(r[0]["allerrs"], exp.allerrs, rtol=0, atol=0)
    np.testing.assert_allclose(r[0]["allfuns"], exp.allfuns, rtol=1e-12)


def _worker_chain(rank, world, n, kw, sub, outdir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    from fake_engine import FakeProblem, FakeSolver
    from oracle import problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.proximal_gradient import NativeRun

    _init_gloo(dist, rank, world, outdir)
    d, c, lam = P.make_pdiag(n, seed=1)
    lo, hi = rank * n // world, (rank + 1) * n // world
    prob = FakeProblem(d[lo:hi], c[lo:hi], lam, group=dist.group.WORLD, world=world, rank=rank)
    base = dict(lr=1, tol=1e-5, tol_internal=1e-12, max_iter=1000000, max_iter_internal=100000,
                max_backtrack_iter=100, warm_start=False, decay_rate=0.5, nesterov=False,
                nesterov_ratio=(0, 0.25), return_all=False, verbose=False, deprecated=False, sub_iters=sub)
    run = NativeRun(prob, np.zeros(hi - lo), base | kw,
                    solver_factory=lambda f, o, p, x0: FakeSolver(f, o, p, x0))
    rows = [np.zeros((0, _lib.ZF_TRACE_COLS))]
    while run.status == _lib.ZF_RUNNING:
        rows.append(run.advance(5))
    ctl = run.solver.ctl
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), x=run.solver.get_x(), nit=ctl.nit, status=ctl.status,
             rows=np.concatenate(rows), lr=ctl.lr, passes=run.solver.passes)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sub", [4, 8])
@pytest.mark.parametrize("kw", [
    dict(lr=0.45, nesterov=True, tol=0.0, max_iter=41),
    dict(lr=4.0, nesterov=True, tol=1e-6, max_iter=500),      # chains break on rejections and on termination
])
def test_sharded_pdiag_world2_chained_passes(tmp_path, kw, sub):
    """The sharded path with chains of `sub` trials per pass: ONE all-gather of sub packs per
    pass (not per iteration); every rank replays the same plan decisions."""
    import torch.multiprocessing as mp

    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import _lib

    n, world = 10007, 2
    mp.spawn(_worker_chain, args=(world, n, kw, sub, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    d, c, lam = P.make_pdiag(n, seed=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n),
                                                 return_all=True, **kw)
    assert int(r[0]["nit"]) == int(r[1]["nit"]) == exp.nit
    assert int(r[0]["status"]) == int(r[1]["status"]) and float(r[0]["lr"]) == float(r[1]["lr"])
    assert np.array_equal(r[0]["rows"], r[1]["rows"]) and int(r[0]["passes"]) == int(r[1]["passes"])
    assert np.array_equal(np.concatenate([r[0]["x"], r[1]["x"]]), exp.x)
    np.testing.assert_allclose(r[0]["rows"][:, _lib.TR_ERR], exp.allerrs, rtol=0, atol=0)
    np.testing.assert_allclose(r[0]["rows"][:, _lib.TR_F], exp.allfuns[1:], rtol=1e-12)
    assert int(r[0]["passes"]) < exp.nit   # fewer exchanges than iterations


def _worker_totals(rank, world, outdir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    from zfista_amd.multiobjective import combine_totals

    _init_gloo(dist, rank, world, outdir)
    vals = np.array([0.1 * (rank + 1), 1e16 if rank == 0 else 1.0, -3.0 + rank, float(rank)])
    got_sum = combine_totals(vals, -1, dist.group.WORLD)
    got_max = combine_totals(vals, 2, dist.group.WORLD)
    np.savez(os.path.join(outdir, f"t{rank}.npz"), s=got_sum, m=got_max)
    dist.barrier()
    dist.destroy_process_group()


def test_multiobjective_totals_exchange_world3(tmp_path):
    """C3: raw totals of a sharded multi-objective reduction are combined in rank order on every
    rank (1e16 + 1 + 1 is order-sensitive: the result must be the rank-ordered one everywhere)."""
    import torch.multiprocessing as mp

    world = 3
    mp.spawn(_worker_totals, args=(world, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"t{k}.npz") for k in range(world)]
    exp_sum = np.array([(0.1 + 0.2) + 0.30000000000000004, (1e16 + 1.0) + 1.0, (-3.0 + -2.0) + -1.0, (0.0 + 1.0) + 2.0])
    for k in range(world):
        assert np.array_equal(r[k]["s"], exp_sum)
        assert np.array_equal(r[k]["m"], np.array([exp_sum[0], exp_sum[1], -1.0, exp_sum[3]]))
