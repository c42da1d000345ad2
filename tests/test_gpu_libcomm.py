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
    assert shard.nit == plain.nit and np.array_equal(shard.x, plain.x)
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
    import socket

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL places one rank per device)")
    script = tmp_path / "two_ranks.py"
    script.write_text(_TWO_RANKS)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, ZF_ROOT=ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
