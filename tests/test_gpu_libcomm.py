"""GPU: RCCL inside the library (csrc/zf_comm.hip, zfista_amd/comm.py).  One GPU is enough for a
1-rank communicator: the real ncclCommInitRank / ncclAllGather calls and the solver's own step
sequence for a sharded vector (trial -> all-gather -> decide, all enqueued by
zf_solver_enqueue_steps) run end to end and must reproduce the unsharded solve bit for bit.  The
2-rank test needs two GPUs and skips otherwise (the driver's multi-GPU runs exercise that path)."""
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_communicator_in_the_library():
    import torch

    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.comm import LibComm
    from zfista_amd.problems import DiagQuadL1, LeastSquaresL1

    comm = LibComm(0, 1, LibComm.new_unique_id())
    a = torch.arange(8, dtype=torch.float64, device="cuda")
    b = torch.zeros(8, dtype=torch.float64, device="cuda")
    comm.all_gather(a, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # what the communicator says about itself (zf_comm_describe): RCCL's OWN queries of the ncclComm_t, the library file
    # the symbols came from, the all-gathers issued - the record bench.py puts on its line (config.rccl)
    info = comm.describe()
    assert info["via"].startswith("zf_comm (RCCL") and (info["world"], info["rank"]) == (1, 0)
    assert info["rank_count_seen"] == 1 and info["rccl_user_rank"] == 0 and info["rccl_device"] == torch.cuda.current_device()
    assert "rccl" in info["library"].lower() and info["rccl_version"] > 0 and info["all_gathers_issued"] == 1
    n = 50001
    d, c, lam = P.make_pdiag(n, seed=1)
    kw = dict(lr=4.0, nesterov=True, tol=1e-8, max_iter=80)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = minimize_proximal_gradient(*DiagQuadL1(d, c, lam).callbacks(), np.zeros(n), **kw)
        shard = minimize_proximal_gradient(*DiagQuadL1(d, c, lam, group=comm).callbacks(), np.zeros(n), **kw)
        exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n), **kw)
    assert shard.nit == plain.nit == exp.nit and np.array_equal(shard.x, plain.x) and shard.fun == plain.fun
    assert np.array_equal(shard.x, exp.x)
    A, bb, lam = P.make_plasso(64, 128, seed=0)
    kw = dict(lr=1.0, nesterov=True, tol=0.0, max_iter=30)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = minimize_proximal_gradient(*LeastSquaresL1(A, bb, lam).callbacks(), np.zeros(128), **kw)
        shard = minimize_proximal_gradient(*LeastSquaresL1(A, bb, lam, group=comm).callbacks(), np.zeros(128), **kw)
    # (the unsharded small matrix takes the two-launch path, the sharded sequence the general one: other
    #  summation orders of grad and f, same iterates to rounding)
    assert shard.nit == plain.nit and np.linalg.norm(shard.x - plain.x) <= 1e-12 * np.linalg.norm(plain.x)
    assert comm.describe()["all_gathers_issued"] > 10      # every pass of the sharded solves exchanged through it
    # the per-pass exchange timed by the solver itself (zf_solver_exchange_stats) and the per-launch records
    from zfista_amd import _lib
    from zfista_amd.proximal_gradient import NativeRun

    o = dict(lr=0.45, tol=0.0, tol_internal=1e-12, max_iter=64, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
             nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=16)
    run = NativeRun(DiagQuadL1(d, c, lam, group=comm), np.zeros(n), o, timing=True)
    while run.status == _lib.ZF_RUNNING:
        run.advance(2)
    ms, cnt = run.solver.exchange_stats()
    rec = run.solver.pass_records()
    assert len(rec) == 4 and cnt >= 4 and 0.0 < ms < 5.0, (ms, cnt, rec)   # (the init exchange is timed too)
    assert all(lag == 0 and fresh == 16 and passes == 0 and t > 0 for lag, fresh, passes, t in rec), rec
    run.solver.close()
    comm.close()


_TWO_RANKS = r"""
import os, sys, warnings
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ZF_ROOT"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
from oracle import cpu_ref, problems_ref as P
from zfista_amd import minimize_proximal_gradient
from zfista_amd.problems import DiagQuadL1
n = 40002
d, c, lam = P.make_pdiag(n, seed=1)
lo, hi = rank * n // world, (rank + 1) * n // world
kw = dict(lr=4.0, nesterov=True, tol=1e-8, max_iter=60)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    res = minimize_proximal_gradient(*DiagQuadL1(d[lo:hi], c[lo:hi], lam, group=dist.group.WORLD).callbacks(), np.zeros(hi - lo), **kw)
    exp = cpu_ref.minimize_proximal_gradient(*P.DiagQuadL1Ref(d, c, lam).callbacks(), np.zeros(n), **kw)
assert res.nit == exp.nit, (res.nit, exp.nit)
assert np.linalg.norm(res.x - exp.x[lo:hi]) <= 1e-10 * np.linalg.norm(exp.x)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_rccl_communicator_in_the_library(tmp_path):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL places one rank per device)")
    script = tmp_path / "two_ranks.py"
    script.write_text(_TWO_RANKS)
    # --standalone: the agent listens on a port the kernel gives it (bind 0, kept open) and the ranks reuse the
    # agent's store - no port number is ever picked, closed and hoped to be still free
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
                          "--nnodes=1", "--nproc-per-node", "2", str(script)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, ZF_ROOT=ROOT))
    assert out.returncode == 0, out.stderr[-3000:]


_FROM_GROUP = r"""
import os, sys, warnings
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ZF_ROOT"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="file://" + os.environ["ZF_RDZV_FILE"], rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
from zfista_amd.comm import LibComm
expect_comm = os.environ.get("ZF_RCCL_LIB") is None
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    comm = LibComm.from_group(None)
    again = LibComm.from_group(None)          # cached verdict, no second collective round
assert again is comm
if expect_comm:
    assert comm is not None and (comm.rank, comm.world) == (0, 1) and not w
    a = torch.arange(8, dtype=torch.float64, device="cuda")
    b = torch.zeros(8, dtype=torch.float64, device="cuda")
    comm.all_gather(a, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
else:
    assert comm is None and len(w) == 1 and "torch.distributed all-gathers" in str(w[0].message), [str(x.message) for x in w]
dist.barrier(); dist.destroy_process_group()
print("ok")
"""


@pytest.mark.parametrize("rccl", ["present", "missing"])
def test_communicator_from_an_nccl_process_group_is_all_or_nothing(tmp_path, rccl):
    """`LibComm.from_group` on a (1-rank) nccl process group: unique id by object broadcast, ncclCommInitRank, and an
    all-reduce of "did every rank get one".  With a librccl that cannot be loaded (ZF_RCCL_LIB names a file that does
    not exist) every rank gets None and a RuntimeWarning - the callers then exchange through torch.distributed -
    instead of an exception on some ranks and a hang on the others."""
    script = tmp_path / "from_group.py"
    script.write_text(_FROM_GROUP)
    env = dict(os.environ, ZF_ROOT=ROOT, ZF_RDZV_FILE=str(tmp_path / "rdzv"))     # file rendezvous: no port
    env.pop("ZF_RCCL_LIB", None)
    if rccl == "missing":
        env["ZF_RCCL_LIB"] = str(tmp_path / "no_such_librccl.so")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("kind", ["diag", "diag_backtrack_history", "lasso", "lasso_rows"])
def test_library_multi_rank_step_sequence_with_thread_ranks(kind, world):
    """The library's OWN multi-rank sequence (zf_solver_enqueue_init_all / zf_solver_enqueue_steps with
    a communicator attached: trial -> all-gather -> decide, rank-major pack layout, rank-ordered sums)
    with `world` ranks on this one GPU: each rank is a host thread with its own stream, the
    communicators are the in-process stand-in (zf_comm_create_local_group) behind the same
    zf_comm_all_gather the RCCL communicator serves.  Against the oracle on the unsharded problem;
    every rank must report bitwise-identical scalars."""
    import threading

    import torch

    from oracle import cpu_ref, problems_ref as P
    from zfista_amd import minimize_proximal_gradient
    from zfista_amd.comm import LibComm
    from zfista_amd.problems import DiagQuadL1, LeastSquaresL1

    if kind.startswith("lasso"):
        m, n = 96, 192
        A, b, lam = P.make_plasso(m, n, seed=2)
        kw = dict(lr=1.0, nesterov=True, tol=0.0, max_iter=25, return_all=True)
        ref = P.LeastSquaresL1Ref(A, b, lam)
    else:
        n = 30011
        d, c, lam = P.make_pdiag(n, seed=1)
        kw = dict(lr=4.0 if "backtrack" in kind else 0.45, nesterov=True, tol=1e-8, max_iter=70,
                  return_all="history" in kind)
        ref = P.DiagQuadL1Ref(d, c, lam)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = cpu_ref.minimize_proximal_gradient(*ref.callbacks(), np.zeros(n), **kw)
    comms = LibComm.local_group(world, cap_doubles=4096)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            lo, hi = r * n // world, (r + 1) * n // world
            with torch.cuda.stream(torch.cuda.Stream()):
                if kind == "lasso_rows":   # rows of A and b split, x replicated
                    r0, r1 = r * m // world, (r + 1) * m // world
                    lo, hi = 0, n
                    prob = LeastSquaresL1(np.ascontiguousarray(A[r0:r1]), b[r0:r1], lam, group=comms[r], shard="rows")
                elif kind == "lasso":
                    prob = LeastSquaresL1(np.ascontiguousarray(A[:, lo:hi]), b, lam, group=comms[r])
                else:
                    prob = DiagQuadL1(d[lo:hi], c[lo:hi], lam, group=comms[r])
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    res = minimize_proximal_gradient(*prob.callbacks(), np.zeros(hi - lo), **kw)
                torch.cuda.current_stream().synchronize()
            out[r] = res
        except Exception as exc:   # pragma: no cover - reported below
            errs.append(exc)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank thread did not finish"
    for res in out:
        assert res.nit == exp.nit == out[0].nit and res.fun == out[0].fun, "ranks must agree bit for bit"
    if kind == "lasso_rows":
        assert all(np.array_equal(o.x, out[0].x) for o in out), "the replicated x must be identical on every rank"
        cat = lambda vs: vs[0]   # noqa: E731
    else:
        cat = np.concatenate
    x = cat([o.x for o in out])
    assert np.linalg.norm(x - exp.x) <= 1e-10 * np.linalg.norm(exp.x)
    np.testing.assert_allclose(out[0].fun, exp.fun, rtol=1e-10)
    if kw.get("return_all"):
        for k in (1, exp.nit // 2, exp.nit):
            xk = cat([o.allvecs[k] for o in out])
            assert np.linalg.norm(xk - exp.allvecs[k]) <= 1e-10 * max(1.0, np.linalg.norm(exp.allvecs[k]))
        np.testing.assert_allclose(out[0].allerrs, exp.allerrs, rtol=1e-10, atol=1e-300)
    for c_ in comms:
        c_.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["jos1_l1", "fds_l1", "fds_box", "jos1_l1_n1e5"])
def test_sharded_multiobjective_native_search_through_the_library_communicator(case, world, golden):
    """SURVEY 8e row 3 + 8f-1 together: x sharded over the ranks of a LIBRARY communicator (zf_mo_set_comm), the
    library's own dual search (dual_solver="native") exchanging ONCE PER BATCH of its state machine - the start
    point with its m curvature probes, the two step lengths of a line search - through zf_comm_all_gather on the
    stream, rank-ordered sums on the device, no Python per exchange.  Thread ranks on this one GPU (the
    in-process stand-in behind the same zf_comm_all_gather RCCL serves).  Against the single-rank solve of the
    same problem (1e-9), against the reference's G4 traces where they exist (the stated 2e-5 of the library's
    search), all ranks bit-identical, and far fewer collectives than dual evaluations."""
    import threading

    import torch

    from zfista_amd.comm import LibComm
    from zfista_amd.problems import FDS, JOS1

    if case == "jos1_l1":
        n, tag, kw0 = 1000, "jos1_n1000_l1", dict(lr=1.0)
        mk = lambda g: JOS1(n, l1_ratios=np.arange(1, 3) / n, l1_shifts=[0, 1], group=g)   # noqa: E731
    elif case == "fds_l1":
        n, tag, kw0 = 100, "fds_n100_l1", dict(lr=1e-3)
        mk = lambda g: FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0, 1, 2], group=g)   # noqa: E731
    elif case == "fds_box":
        n, tag, kw0 = 103, None, dict(lr=1e-3)
        mk = lambda g: FDS(n, bounds=(-1.5, 1.8), group=g)   # noqa: E731
    else:
        n, tag, kw0 = 100003, None, dict(lr=0.4 * 100003)
        mk = lambda g: JOS1(n, l1_ratios=np.arange(1, 3) / n, l1_shifts=[0, 1], group=g)   # noqa: E731
    kw = dict(nesterov=True, tol=1e-5, max_iter=12, return_all=True, dual_solver="native", **kw0)
    G = golden("g4_multiobjective.npz")
    x0 = G(f"{tag}.x0") if tag else np.random.default_rng(3).uniform(-1, 1, n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = mk(None).minimize_proximal_gradient(x0, **kw)
    comms = LibComm.local_group(world, cap_doubles=4096)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                prob = mk(comms[r])
                lo, hi = prob.shard_bounds()
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    res = prob.minimize_proximal_gradient(x0[lo:hi], **kw)
                torch.cuda.current_stream().synchronize()
                eng = prob._engine()
                out[r] = (res, eng.exchange_count(), eng.n_dual_evals, eng.n_exchanges)
        except Exception as exc:   # pragma: no cover - reported below
            errs.append(exc)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank thread did not finish"
    res0 = out[0][0]
    for res, n_coll, n_dual, n_py in out:
        assert res.nit == res0.nit == full.nit and res.status == full.status
        assert np.array_equal(np.asarray(res.allerrs), np.asarray(res0.allerrs)), "ranks must agree bit for bit"
        assert np.array_equal(np.stack(res.allfuns), np.stack(res0.allfuns))
        assert n_py == 0, "no exchange may go through the Python callback"
        # one collective per BATCH of the search (+ f, g, recovery per trial): well below one per evaluation
        assert n_dual > 0 and n_coll < n_dual + 6 * res.nit + 8, (n_coll, n_dual)
    m = 2 if case.startswith("jos1") else 3
    if m == 3:   # the curvature probes travel with their point: at most ~half as many collectives as evaluations
        assert out[0][1] - 6 * res0.nit - 8 <= 0.75 * out[0][2], (out[0][1], out[0][2])
    for k in range(res0.nit + 1):
        xk = np.concatenate([o[0].allvecs[k] for o in out])
        assert np.linalg.norm(xk - full.allvecs[k]) <= 1e-9 * max(1.0, np.linalg.norm(full.allvecs[k])), k
        if tag:   # the reference's own run of this case (G4), at the accuracy of the library's search
            ref = G(f"{tag}.fista.vecs")[k]
            assert np.linalg.norm(xk - ref) <= 2e-5 * max(1.0, np.linalg.norm(ref)), k
    np.testing.assert_allclose(np.stack(res0.allfuns), np.stack(full.allfuns), rtol=1e-9)
    if tag:
        assert res0.nit == int(G(f"{tag}.fista.nit"))
    for c_ in comms:
        c_.close()


def test_kernel_timing_of_a_row_sharded_solver_with_thread_ranks():
    """zf_solver_set_timing on a ROW-sharded least-squares solver: the prox step of a trial runs after the exchange
    (zf_solver_enqueue_trial_finish), so the first half of the trial must not reserve an event pair it never
    records (zf_collect_timing then read unrecorded events: an error, or garbage).  Two rank threads; the timing
    calls must succeed and report sane numbers; the cache-resident two-launch path logs its passes as well."""
    import threading

    import torch

    from oracle import problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.comm import LibComm
    from zfista_amd.problems import LeastSquaresL1
    from zfista_amd.proximal_gradient import NativeRun

    m, n, world = 96, 192, 2
    A, b, lam = P.make_plasso(m, n, seed=2)
    o = dict(lr=1.0, tol=0.0, tol_internal=1e-12, max_iter=12, max_backtrack_iter=100, decay_rate=0.5, nesterov=True,
             nesterov_ratio=(0, 0.25), deprecated=False)
    comms = LibComm.local_group(world, cap_doubles=4096)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            r0, r1 = r * m // world, (r + 1) * m // world
            with torch.cuda.stream(torch.cuda.Stream()):
                prob = LeastSquaresL1(np.ascontiguousarray(A[r0:r1]), b[r0:r1], lam, group=comms[r], shard="rows")
                run = NativeRun(prob, np.zeros(n), o, timing=True)
                while run.status == _lib.ZF_RUNNING:
                    run.advance(3)
                out[r] = (run.nit_seen, run.solver.trial_kernel_ms())
                run.solver.close()
        except Exception as exc:   # pragma: no cover - reported below
            errs.append(exc)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs
    for nit, (ms, cnt) in out:
        assert nit == 12 and ms >= 0.0 and cnt >= 0
    for c_ in comms:
        c_.close()
    # the cache-resident two-launch path (unsharded): its passes are logged, full-chain statistics are consistent
    prob = LeastSquaresL1(A, b, lam)
    run = NativeRun(prob, np.zeros(n), o, timing=True)
    while run.status == _lib.ZF_RUNNING:
        run.advance(3)
    (fm, fn), (pm, pn) = run.solver.pass_stats()
    assert fn + pn >= 12 and fm > 0.0   # (S = 1: every trial is a "full chain" of one)
    run.solver.close()


@pytest.mark.parametrize("world", [1, 3])
def test_a_sharded_pass_launches_only_the_kernel_of_its_shape(world):
    """A chained pass has several shape-specific kernels and needs one.  Once a poll has shown the host the control
    block it launches only the one it expects - through the library's communicator too: every rank predicts from the
    same control block, the packs carry a stamp of the state they were computed from (zf_pack_stamp) and the decide
    step ignores gathered packs of any other state.  Checked here: the launch counts, and bitwise agreement with
    the solve that launches every kernel every time (ZF_SPECULATE=0)."""
    import threading

    import torch

    from oracle import problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.comm import LibComm
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    n, K = 3 * 40000 + 17, 16 * 3 + 20   # three full chains, then a shared tail of 10 + 10 (no rejections yet)
    d, c, lam = P.make_pdiag(n, seed=3)
    opts = dict(lr=0.45, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5,
                nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=16)
    dd, cc = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
    x0 = torch.zeros(n, dtype=torch.float64, device="cuda")

    def solve(prob, x0, chunk):
        run = NativeRun(prob, x0, opts)
        counts, rows = [], []
        while run.status == _lib.ZF_RUNNING:
            rows.append(run.advance(chunk))
            counts.append(run.solver.launch_counts())
        x = run.solver.get_x()
        nit = int(run.solver.ctl.nit)
        run.solver.close()
        return np.concatenate(rows), x, nit, counts

    def sharded(chunk):
        comms = LibComm.local_group(world, cap_doubles=4096) if world > 1 else [LibComm(0, 1, LibComm.new_unique_id())]
        out, errs = [None] * world, []

        def rank_main(r):
            try:
                lo, hi = r * n // world, (r + 1) * n // world
                with torch.cuda.stream(torch.cuda.Stream()):
                    out[r] = solve(DiagQuadL1(dd[lo:hi].clone(), cc[lo:hi].clone(), lam, group=comms[r]), x0[lo:hi].clone(), chunk)
                    torch.cuda.current_stream().synchronize()
            except Exception as exc:   # pragma: no cover - reported below
                errs.append(exc)

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errs, errs
        assert all(o is not None for o in out), "a rank thread did not finish"
        for c_ in comms:
            c_.close()
        return out

    os.environ["ZF_SPECULATE"] = "0"   # (read when a solver is created)
    try:
        everything = sharded(2)    # every kernel of every pass
    finally:
        del os.environ["ZF_SPECULATE"]
    for rows, x, nit, counts in everything:
        assert nit == K and counts[-1] == (6, 18), counts   # (the three kernels that between them run every shape)
    by_chunk = sharded(2)          # polled every two passes
    for r, (rows, x, nit, counts) in enumerate(by_chunk):
        assert nit == K
        assert np.array_equal(rows, everything[r][0]) and np.array_equal(x, everything[r][1])
        # 3 full chains + 10 + 10 = 5 passes in 3 chunks of 2 steps: one kernel each; the 6th step finds the solve
        # finished, as the host expected - then it launches the general body on any shape, in case the device is NOT
        assert np.all(rows[:, _lib.TR_TRIALS] == 1), "the test wants a solve without rejections"
        assert counts[-1] == (6, 5 + 1), counts


@pytest.mark.parametrize("world", [1, 2])
def test_after_rejections_a_pass_launches_the_two_shapes_it_can_need(world):
    """Once a poll has seen rejections the host no longer knows which shape the passes behind the first have: the
    general body runs them whatever their shape (on this small grid alone; on large ones beside the expected kernel).
    A solve that starts with lr too large (the first line search backtracks) and runs into the noise floor of the
    acceptance test (rejections from iteration ~60 on at this size), polled every 2 passes, against the same solve
    with every kernel launched every time (ZF_SPECULATE=0): identical rows and iterates, fewer launches."""
    import threading

    import torch

    from oracle import problems_ref as P
    from zfista_amd import _lib
    from zfista_amd.comm import LibComm
    from zfista_amd.problems import DiagQuadL1
    from zfista_amd.proximal_gradient import NativeRun

    n, K = 2 * 60000 + 18, 260
    d, c, lam = P.make_pdiag(n, seed=3)
    opts = dict(lr=4.0, tol=0.0, tol_internal=1e-12, max_iter=K, max_backtrack_iter=100, decay_rate=0.5,
                nesterov=True, nesterov_ratio=(0, 0.25), deprecated=False, sub_iters=16)
    dd, cc = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
    x0 = torch.zeros(n, dtype=torch.float64, device="cuda")

    def solve(prob, x0):
        run = NativeRun(prob, x0, opts)
        rows = []
        while run.status == _lib.ZF_RUNNING:
            rows.append(run.advance(2))
        x, nit, counts = run.solver.get_x(), int(run.solver.ctl.nit), run.solver.launch_counts()
        run.solver.close()
        return np.concatenate(rows), x, nit, counts

    def ranks():
        comms = LibComm.local_group(world, cap_doubles=4096) if world > 1 else [LibComm(0, 1, LibComm.new_unique_id())]
        out, errs = [None] * world, []

        def rank_main(r):
            try:
                lo, hi = r * n // world, (r + 1) * n // world
                with torch.cuda.stream(torch.cuda.Stream()):
                    out[r] = solve(DiagQuadL1(dd[lo:hi].clone(), cc[lo:hi].clone(), lam, group=comms[r]), x0[lo:hi].clone())
                    torch.cuda.current_stream().synchronize()
            except Exception as exc:   # pragma: no cover - reported below
                errs.append(exc)

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errs, errs
        assert all(o is not None for o in out), "a rank thread did not finish"
        for c_ in comms:
            c_.close()
        return out

    os.environ["ZF_SPECULATE"] = "0"
    try:
        everything = ranks()
    finally:
        del os.environ["ZF_SPECULATE"]
    predicted = ranks()
    for r in range(world):
        rows_e, x_e, nit_e, (steps_e, kernels_e) = everything[r]
        rows_p, x_p, nit_p, (steps_p, kernels_p) = predicted[r]
        assert nit_e == nit_p == K
        assert np.any(rows_e[:, _lib.TR_TRIALS] > 1), "the test wants rejections"
        assert np.array_equal(rows_e, rows_p) and np.array_equal(x_e, x_p)
        assert kernels_e == 3 * steps_e
        # mispredicted steps are no-ops that cost a step each: a few, and far fewer kernels all the same
        assert steps_e <= steps_p <= steps_e + 8, (steps_e, steps_p)
        assert kernels_p <= 0.7 * kernels_e, (kernels_p, kernels_e)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["jos1_l1", "fds_l1", "fds_box", "jos1_l1_n1e5"])
def test_sharded_multiobjective_device_driven_search(case, world, golden):
    """Round 5 (SURVEY 8e row 3 + 8f-1): dual_solver="device" with x sharded over a library communicator.  The persistent
    kernel of the single-rank device search cannot exchange between ranks; here the state machine lives in device memory
    and a batch is evaluation -> reduce -> ONE all-gather -> a one-wave kernel that advances it (zf_mo_solve_dual_stream):
    no host synchronisation per batch.  Thread ranks on this GPU.  Against the single-rank DEVICE solve of the same problem
    (1e-9: the probing search here, the exact-Hessian search there - both converge to the optimum of the dual), against
    the host-driven search of the same sharded problem (the same machine on the same numbers: bit for bit), all ranks
    bit-identical, `dual_search_trials` says which search ran."""
    import threading

    import torch

    from zfista_amd.comm import LibComm
    from zfista_amd.problems import FDS, JOS1

    if case == "jos1_l1":
        n, tag, kw0 = 1000, "jos1_n1000_l1", dict(lr=1.0)
        mk = lambda g: JOS1(n, l1_ratios=np.arange(1, 3) / n, l1_shifts=[0, 1], group=g)   # noqa: E731
    elif case == "fds_l1":
        n, tag, kw0 = 100, "fds_n100_l1", dict(lr=1e-3)
        mk = lambda g: FDS(n, l1_ratios=np.arange(1, 4) / n, l1_shifts=[0, 1, 2], group=g)   # noqa: E731
    elif case == "fds_box":
        n, tag, kw0 = 103, None, dict(lr=1e-3)
        mk = lambda g: FDS(n, bounds=(-1.5, 1.8), group=g)   # noqa: E731
    else:
        n, tag, kw0 = 100003, None, dict(lr=0.4 * 100003)
        mk = lambda g: JOS1(n, l1_ratios=np.arange(1, 3) / n, l1_shifts=[0, 1], group=g)   # noqa: E731
    kw = dict(nesterov=True, tol=1e-5, max_iter=12, return_all=True, **kw0)
    G = golden("g4_multiobjective.npz")
    x0 = G(f"{tag}.x0") if tag else np.random.default_rng(3).uniform(-1, 1, n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = mk(None).minimize_proximal_gradient(x0, dual_solver="device", **kw)

    def sharded(solver):
        comms = LibComm.local_group(world, cap_doubles=4096)
        out, errs = [None] * world, []

        def rank_main(r):
            try:
                with torch.cuda.stream(torch.cuda.Stream()):
                    prob = mk(comms[r])
                    lo, hi = prob.shard_bounds()
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        res = prob.minimize_proximal_gradient(x0[lo:hi], dual_solver=solver, **kw)
                    torch.cuda.current_stream().synchronize()
                    eng = prob._engine()
                    out[r] = (res, eng.exchange_count(), eng.n_dual_evals, eng.n_exchanges)
            except Exception as exc:   # pragma: no cover - reported below
                errs.append(exc)

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        for c_ in comms:
            c_.close()
        assert not errs, errs
        assert all(o is not None for o in out), "a rank thread did not finish"
        return out

    dev = sharded("device")
    nat = sharded("native")
    res0 = dev[0][0]
    for (res, n_coll, n_dual, n_py), (resn, *_) in zip(dev, nat):
        assert res.nit == res0.nit == full.nit and res.status == full.status
        assert res["dual_search_trials"]["device"] >= res.nit and res["dual_search_trials"]["native"] == 0
        assert np.array_equal(np.asarray(res.allerrs), np.asarray(res0.allerrs)), "ranks must agree bit for bit"
        assert np.array_equal(np.stack(res.allfuns), np.stack(res0.allfuns))
        assert n_py == 0 and n_dual > 0
        # the same machine on the same numbers as the host-driven search of the sharded problem
        assert resn.nit == res.nit and np.array_equal(np.stack(resn.allfuns), np.stack(res.allfuns))
        assert all(np.array_equal(a, b) for a, b in zip(resn.allvecs, res.allvecs))
    for k in range(res0.nit + 1):
        xk = np.concatenate([o[0].allvecs[k] for o in dev])
        assert np.linalg.norm(xk - full.allvecs[k]) <= 1e-9 * max(1.0, np.linalg.norm(full.allvecs[k])), k
    np.testing.assert_allclose(np.stack(res0.allfuns), np.stack(full.allfuns), rtol=1e-9)
