"""BASELINE cfg5 at FULL size on one GPU: the decision vector of n = 8 x 10^8 sharded over 8 ranks
(contiguous blocks of x, d, c; one packed all-gather per pass; rank-ordered sums)
against the SAME problem solved unsharded by one solver.

No 8-GPU node is available to the build, but all eight 10^8-element shards fit the 288 GB of one
MI355X: every rank is a host thread with its own stream and its own ``zf_solver``; the communicators
are the library's in-process group (``zf_comm_create_local_group``) behind the same
``zf_comm_all_gather`` an RCCL communicator serves, so the library's own multi-rank step sequence
(``zf_solver_enqueue_steps`` with a communicator attached), the pack layout and the decide pass run
exactly as they would with one rank per GPU - only the transport differs.  This is SURVEY 8d's check
"at scale by 1-GPU-vs-8-GPU agreement and scalar traces"."""
from __future__ import annotations

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORLD = 8
N_SHARD = 10**8
LAM = 0.1


@pytest.mark.parametrize("lr,K", [(0.45, 48), (4.0, 20)])   # the bench's clean regime; and with backtracking
def test_cfg5_sharded_8x1e8_equals_unsharded(lr, K):
    import torch

    from zfista_amd import _lib
    from zfista_amd.comm import LibComm
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    free, _ = torch.cuda.mem_get_info()
    if free < 110 * 2**30:
        pytest.skip("needs ~100 GB of free HBM")
    n = WORLD * N_SHARD
    gen = torch.Generator(device="cuda").manual_seed(5)
    d = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 1.5 + 0.5
    c = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
    opts = dict(lr=lr, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5,
                nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=0)

    def solve(prob, x0):
        run = NativeRun(prob, x0, opts)
        rows = []
        while run.status == _lib.ZF_RUNNING:
            rows.append(run.advance(4))
        x = run.solver.get_x()
        nit, status, F = int(run.solver.ctl.nit), run.status, float(run.solver.ctl.F_old)
        run.solver.close()
        return np.concatenate(rows), x, nit, status, F

    x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    rows1, x1, nit1, status1, F1 = solve(DiagQuadL1(d, c, LAM), x0)
    assert nit1 == K and status1 == _lib.ZF_MAXITER and len(rows1) == K
    if lr > 1.0:
        assert rows1[0, _lib.TR_LR] < lr, "the first line search must have backtracked"

    comms = LibComm.local_group(WORLD, cap_doubles=4096)
    out, errs = [None] * WORLD, []

    def rank_main(r):
        try:
            lo, hi = r * N_SHARD, (r + 1) * N_SHARD
            with torch.cuda.stream(torch.cuda.Stream()):
                out[r] = solve(DiagQuadL1(d[lo:hi], c[lo:hi], LAM, group=comms[r]), x0[lo:hi])
                torch.cuda.current_stream().synchronize()
        except Exception as exc:   # pragma: no cover - reported below
            errs.append(exc)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank thread did not finish"
    rows0 = out[0][0]
    for r, (rows, x, nit, status, F) in enumerate(out):
        # every rank adds the gathered packs in rank order: bitwise-identical scalars and decisions
        assert nit == K and status == status1 and F == out[0][4]
        assert np.array_equal(rows, rows0), f"rank {r} took other decisions than rank 0"
        # the iterate arithmetic is elementwise: with equal decisions the shard IS the slice, bit for bit
        assert np.array_equal(x, x1[r * N_SHARD:(r + 1) * N_SHARD]), f"shard {r} differs from the unsharded solve"
    # scalar traces: the same sums in another order (8 rank totals instead of one): 1e-10, as everywhere
    for col in (_lib.TR_ERR, _lib.TR_F, _lib.TR_LR):
        np.testing.assert_allclose(rows0[:, col], rows1[:, col], rtol=1e-10, atol=0)
    # the model value (:149-155) is f(y) - F(x_k) + ... : a difference of O(|F|) sums, resolved to ulp(F)
    np.testing.assert_allclose(rows0[:, _lib.TR_FUN], rows1[:, _lib.TR_FUN], rtol=1e-10,
                               atol=1e-10 * np.max(np.abs(rows1[:, _lib.TR_F])))
    assert np.array_equal(rows0[:, _lib.TR_LR], rows1[:, _lib.TR_LR])   # the same step sizes exactly
    assert abs(out[0][4] - F1) <= 1e-10 * abs(F1)
    for c_ in comms:
        c_.close()
