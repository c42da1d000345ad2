"""GPU: the sharded operators in REAL separate processes (one per rank, torch.distributed over
gloo, all ranks on this one GPU - RCCL itself needs a device per rank and is exercised with one
rank by tools/check_sharded_mo_rccl.py and ``ZF_FORCE_SPLIT=1 torchrun ... bench.py``).  Complements
the in-process lockstep tests of test_gpu_sharded.py: here the exchange is a real collective between
processes and every rank drives its own solver through the public entry point."""
import os
import sys
import warnings

import numpy as np
import pytest

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, case, outdir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    from oracle import problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.problems import DiagQuadL1, LeastSquaresL1

    torch.cuda.set_device(0)
    # file rendezvous: no port is picked, so none can be taken by somebody else before rank 0 listens on it
    dist.init_process_group("gloo", init_method=f"file://{outdir}/rdzv", rank=rank, world_size=world)
    warnings.simplefilter("ignore")
    if case == "lasso_rows":   # rows of A and b split over the ranks, x replicated: A_p^T r_p is exchanged
        A, b, lam = P.make_plasso(96, 301, seed=3, n_informative=12)
        r0, r1 = rank * 96 // world, (rank + 1) * 96 // world
        prob = LeastSquaresL1(np.ascontiguousarray(A[r0:r1]), b[r0:r1], lam, group=dist.group.WORLD, shard="rows")
        kw = dict(lr=1.0, nesterov=True, tol=1e-8, max_iter=60)
        n_loc = 301
    elif case == "lasso":
        A, b, lam = P.make_plasso(96, 301, seed=3, n_informative=12)
        lo, hi = rank * 301 // world, (rank + 1) * 301 // world
        prob = LeastSquaresL1(np.ascontiguousarray(A[:, lo:hi]), b, lam, group=dist.group.WORLD)
        kw = dict(lr=1.0, nesterov=True, tol=1e-8, max_iter=60)
        n_loc = hi - lo
    else:
        n = 50021
        d, c, lam = P.make_pdiag(n, seed=9)
        lo, hi = rank * n // world, (rank + 1) * n // world
        prob = DiagQuadL1(d[lo:hi], c[lo:hi], lam, group=dist.group.WORLD)
        kw = dict(lr=8.0, nesterov=True, tol=1e-9, max_iter=120)     # backtracking, chains of 8, termination
        n_loc = hi - lo
    res = minimize_proximal_gradient(*prob.callbacks(), np.zeros(n_loc), **kw)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), x=res.x, nit=res.nit, status=res.status, fun=res.fun)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("lasso", 2), ("lasso", 3), ("lasso_rows", 2), ("lasso_rows", 3), ("diag", 2),
                                        ("diag", 4)])
def test_sharded_solve_in_separate_processes(case, world, tmp_path):
    import torch.multiprocessing as mp

    from oracle import cpu_ref, problems_ref as P

    mp.spawn(_worker, args=(world, case, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if case.startswith("lasso"):
            A, b, lam = P.make_plasso(96, 301, seed=3, n_informative=12)
            exp = cpu_ref.minimize_proximal_gradient(*P.LeastSquaresL1Ref(A, b, lam).callbacks(), np.zeros(301), lr=1.0,
                                                     nesterov=True, tol=1e-8, max_iter=60)
        else:
            d, c, lam = P.make_pdiag(50021, seed=9)
            exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(50021), lr=8.0,
                                                     nesterov=True, tol=1e-9, max_iter=120)
    assert all(int(k["nit"]) == exp.nit and int(k["status"]) == exp.status for k in r)
    assert all(float(k["fun"]) == float(r[0]["fun"]) for k in r), "ranks must agree bit for bit"
    if case == "lasso_rows":
        assert all(np.array_equal(k["x"], r[0]["x"]) for k in r), "the replicated x must be identical on every rank"
        assert rel_err(r[0]["x"], exp.x) <= 1e-10
    else:
        assert rel_err(np.concatenate([k["x"] for k in r]), exp.x) <= 1e-10
    np.testing.assert_allclose(float(r[0]["fun"]), exp.fun, rtol=1e-10)
